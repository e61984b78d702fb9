//! Pins the oracle with the REFERENCE ITSELF (VERDICT r1 "missing" #2).
//!
//! Links the real `gym-rs` crate, replays the INPUTS of `tests/golden/cartpole.json` / `mountain_car.json` through the
//! reference's own `CartPoleEnv::step` / `MountainCarEnv::step` (cartpole.rs:398-483, mountain_car.rs:398-435), and
//! writes files with the same schema whose OUTPUT fields (`next`, `reward`, `done`, `steps`, `final`, `total_reward`,
//! `rewards`, `dones`) come from the reference.  Plus `reset_kat.json`: the states `reset(Some(seed))` gives for a few
//! seeds (PCG64 + rand's Uniform: cartpole.rs:485-516, seeding.rs:21-26) -- what the optional PCG64 reset mode
//! (`gymrs_reset_pcg64`) reproduces; the default reset draws from Philox by design.
//!
//!     cargo run --release --bin make_golden -- ../../tests/golden ../../tests/golden/from_reference
//!     python -m pytest tests/test_oracle_reference_pins.py tests/test_pcg64_reset.py   # hold the C oracle to from_reference/
//!
//! The image this repository is built in has no cargo/rustc, so this file has never been compiled there; it uses only
//! the crate's public API (pub fields `state`, `kinematics_integrator`; `Env::step`; `Env::reset`).
use gym_rs::core::Env;
use gym_rs::envs::classical_control::cartpole::{CartPoleEnv, CartPoleObservation, KinematicsIntegrator};
use gym_rs::envs::classical_control::mountain_car::{MountainCarEnv, MountainCarObservation};
use gym_rs::utils::renderer::RenderMode;
use ordered_float::OrderedFloat;
use serde_json::{json, Value};
use std::{env, fs, path::Path};

fn f64s(v: &Value) -> Vec<f64> {
    v.as_array().expect("array of numbers").iter().map(|x| x.as_f64().expect("number")).collect()
}

fn cartpole_obs(s: &[f64]) -> CartPoleObservation {
    CartPoleObservation::new(OrderedFloat(s[0]), OrderedFloat(s[1]), OrderedFloat(s[2]), OrderedFloat(s[3]))
}

fn cartpole_vec(o: CartPoleObservation) -> Vec<f64> {
    o.into() // Into<Vec<f64>>: x, x_dot, theta, theta_dot (cartpole.rs:336-349)
}

/// A fresh reference env placed in `state` (the pub `state` field) with `steps_beyond_terminated` cleared.
fn cartpole_at(state: &[f64], integrator: KinematicsIntegrator) -> CartPoleEnv {
    let mut env = CartPoleEnv::new(RenderMode::None);
    env.state = cartpole_obs(state);
    env.steps_beyond_terminated = None;
    env.kinematics_integrator = integrator;
    env
}

fn cartpole_single(cases: &Value, integrator: fn() -> KinematicsIntegrator) -> Value {
    let mut out = Vec::new();
    for case in cases.as_array().expect("list of cases") {
        let state = f64s(&case["state"]);
        let action = case["action"].as_u64().expect("action") as usize;
        let mut env = cartpole_at(&state, integrator());
        let r = env.step(action);
        out.push(json!({"state": state, "action": action, "next": cartpole_vec(r.observation),
                        "reward": r.reward.into_inner(), "done": r.done}));
    }
    Value::Array(out)
}

fn cartpole_policy(name: &str, t: usize) -> usize {
    match name {
        "always_1" => 1,
        "always_0" => 0,
        "alternate_1_0" => (t + 1) % 2,
        other => panic!("unknown policy {other}"),
    }
}

fn cartpole(src: &Value) -> Value {
    let mut trajectories = Vec::new();
    for tr in src["trajectories"].as_array().expect("trajectories") {
        let policy = tr["policy"].as_str().expect("policy");
        let start = f64s(&tr["start"]);
        let mut env = cartpole_at(&start, KinematicsIntegrator::Euler);
        let (mut t, mut total) = (0usize, 0.0f64);
        loop {
            let r = env.step(cartpole_policy(policy, t));
            total += r.reward.into_inner();
            t += 1;
            if r.done {
                break;
            }
        }
        trajectories.push(json!({"policy": policy, "start": start, "steps": t, "final": cartpole_vec(env.state), "total_reward": total}));
    }
    let bt = &src["beyond_terminated"];
    let start = f64s(&bt["start"]);
    let action = bt["action"].as_u64().expect("action") as usize;
    let n = bt["rewards"].as_array().expect("rewards").len();
    let mut env = cartpole_at(&start, KinematicsIntegrator::Euler);
    let (mut rewards, mut dones) = (Vec::new(), Vec::new());
    for _ in 0..n {
        let r = env.step(action); // cartpole.rs:455-464: 1.0 on the terminating step, then 0.0 (and a warning)
        rewards.push(r.reward.into_inner());
        dones.push(r.done);
    }
    let high: Vec<f64> = cartpole_vec(env.observation_space.high);
    json!({
        "constants": {"gravity": env.gravity.into_inner(), "masscart": env.masscart.into_inner(), "masspole": env.masspole.into_inner(),
                      "length": env.length.into_inner(), "force_mag": env.force_mag.into_inner(), "tau": env.tau.into_inner(),
                      "theta_threshold_radians": env.theta_threshold_radians.into_inner(), "x_threshold": env.x_threshold.into_inner()},
        "single_steps": cartpole_single(&src["single_steps"], || KinematicsIntegrator::Euler),
        "trajectories": trajectories,
        "beyond_terminated": {"start": start, "action": action, "rewards": rewards, "dones": dones},
        "semi_implicit": cartpole_single(&src["semi_implicit"], || KinematicsIntegrator::Other),
        // +-inf components print as null in JSON: keep the four numbers as the existing fixture does
        "observation_space_high": high.iter().map(|v| if v.is_finite() { json!(v) } else { json!(null) }).collect::<Vec<_>>(),
    })
}

fn mountain_car_at(state: &[f64]) -> MountainCarEnv {
    let mut env = MountainCarEnv::new(RenderMode::None);
    env.state = MountainCarObservation::new(OrderedFloat(state[0]), OrderedFloat(state[1]));
    env
}

fn mountain_car(src: &Value) -> Value {
    let mut single = Vec::new();
    for case in src["single_steps"].as_array().expect("single_steps") {
        let state = f64s(&case["state"]);
        let action = case["action"].as_u64().expect("action") as usize;
        let mut env = mountain_car_at(&state);
        let r = env.step(action);
        single.push(json!({"state": state, "action": action,
                           "next": [r.observation.position.into_inner(), r.observation.velocity.into_inner()],
                           "reward": r.reward.into_inner(), "done": r.done}));
    }
    let mut trajectories = Vec::new();
    for tr in src["trajectories"].as_array().expect("trajectories") {
        assert_eq!(tr["policy"].as_str(), Some("bang_bang")); // push in the direction of motion (right when at rest)
        let start = f64s(&tr["start"]);
        let mut env = mountain_car_at(&start);
        let (mut t, mut total) = (0usize, 0.0f64);
        loop {
            let action = if env.state.velocity.into_inner() >= 0.0 { 2 } else { 0 };
            let r = env.step(action);
            total += r.reward.into_inner();
            t += 1;
            if r.done || t >= 100_000 {
                break;
            }
        }
        trajectories.push(json!({"policy": "bang_bang", "start": start, "steps": t,
                                 "final": [env.state.position.into_inner(), env.state.velocity.into_inner()], "total_reward": total}));
    }
    let env = MountainCarEnv::new(RenderMode::None);
    json!({
        "constants": {"min_position": env.min_position.into_inner(), "max_position": env.max_position.into_inner(),
                      "max_speed": env.max_speed.into_inner(), "goal_position": env.goal_position.into_inner(),
                      "goal_velocity": env.goal_velocity.into_inner(), "force": env.force.into_inner(), "gravity": env.gravity.into_inner()},
        "single_steps": single,
        "trajectories": trajectories,
    })
}

/// `reset(Some(seed))` of the reference: PCG64 seeded with `seed_from_u64`, four (one) uniform draws.
fn reset_kat() -> Value {
    let seeds = [0u64, 1, 42, 2024, u64::MAX];
    let mut cp = Vec::new();
    let mut mc = Vec::new();
    for seed in seeds {
        let mut env = CartPoleEnv::new(RenderMode::None);
        let (obs, _) = env.reset(Some(seed), false, None);
        cp.push(json!({"seed": seed, "state": cartpole_vec(obs)}));
        let mut env = MountainCarEnv::new(RenderMode::None);
        let (obs, _) = env.reset(Some(seed), false, None);
        mc.push(json!({"seed": seed, "state": [obs.position.into_inner(), obs.velocity.into_inner()]}));
    }
    json!({"cartpole": cp, "mountain_car": mc, "generator": "rand_pcg::Pcg64::seed_from_u64 + rand::distributions::Uniform (cartpole.rs:485-516)"})
}

fn main() {
    let args: Vec<String> = env::args().collect();
    let src_dir = Path::new(args.get(1).map(String::as_str).unwrap_or("../../tests/golden"));
    let out_dir = Path::new(args.get(2).map(String::as_str).unwrap_or("../../tests/golden/from_reference"));
    fs::create_dir_all(out_dir).expect("create the output directory");
    let read = |name: &str| -> Value {
        serde_json::from_str(&fs::read_to_string(src_dir.join(name)).unwrap_or_else(|e| panic!("{name}: {e}"))).expect("valid JSON")
    };
    let write = |name: &str, v: &Value| fs::write(out_dir.join(name), serde_json::to_string(v).expect("serialise")).expect("write");
    write("cartpole.json", &cartpole(&read("cartpole.json")));
    write("mountain_car.json", &mountain_car(&read("mountain_car.json")));
    write("reset_kat.json", &reset_kat());
    println!("wrote cartpole.json, mountain_car.json, reset_kat.json to {}", out_dir.display());
}
