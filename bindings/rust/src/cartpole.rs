//! `CartPoleEnv` with the reference's public surface (cartpole.rs:51-87, 389-560), one GPU lane behind it.
//!
//! The pub physics fields can be edited between steps like the reference's: an edit is noticed on the next
//! `step`/`reset` and pushed down with `gymrs_set_params`, which changes nothing but the constants.
use crate::engine::Engine;
use crate::ffi::CartPoleParams;
use gym_rs::core::{ActionReward, Env, EnvProperties};
use gym_rs::envs::classical_control::cartpole::{CartPoleObservation, KinematicsIntegrator};
use gym_rs::spaces::{BoxR, Discrete, Space};
use gym_rs::utils::custom::structs::Metadata;
use gym_rs::utils::custom::types::O64;
use gym_rs::utils::renderer::{RenderMode, Renders};
use gym_rs::utils::seeding::rand_random;
use ordered_float::OrderedFloat;
use rand_pcg::Pcg64;
use serde::Serialize;

/// GPU-backed drop-in for `gym_rs::envs::classical_control::cartpole::CartPoleEnv`.
#[derive(Debug, Clone, Serialize)]
pub struct CartPoleEnv {
    /// The available actions that can be taken: `Discrete(2)`.
    pub action_space: Discrete,
    /// The range of values that can be observed.
    pub observation_space: BoxR<CartPoleObservation>,
    /// Always `RenderMode::None`: rendering is out of scope of the GPU path.
    pub render_mode: RenderMode,
    /// The current state (host copy, refreshed after every step and reset).
    pub state: CartPoleObservation,
    /// Additional pieces of information provided by the environment.
    pub metadata: Metadata<Self>,
    /// Gravity constant.
    pub gravity: O64,
    /// Mass of the cart.
    pub masscart: O64,
    /// Mass of the pole.
    pub masspole: O64,
    /// Half the length of the pole.
    pub length: O64,
    /// Force applied per action.
    pub force_mag: O64,
    /// Seconds between state updates.
    pub tau: O64,
    /// Euler or semi-implicit Euler.
    pub kinematics_integrator: KinematicsIntegrator,
    /// Pole angle at which an episode terminates.
    pub theta_threshold_radians: O64,
    /// Cart position at which an episode terminates.
    pub x_threshold: O64,
    /// Steps taken after the episode terminated.
    pub steps_beyond_terminated: Option<usize>,
    #[serde(skip_serializing)]
    pushed: CartPoleParams,
    #[serde(skip_serializing)]
    engine: Engine,
    /// `EnvProperties::rand_random` must hand out a `&Pcg64` (core.rs:73).  The device generator is
    /// counter-based Philox, so this one only mirrors the seed: the one source-level deviation.
    #[serde(skip_serializing)]
    rand_random: Pcg64,
}

/// What the reference's metadata lists (cartpole.rs:265); only `RenderMode::None` is honoured here.
const RENDER_MODES: &[RenderMode] = &[RenderMode::Human, RenderMode::RgbArray];

fn observation(st: &[f32]) -> CartPoleObservation {
    CartPoleObservation::new(
        OrderedFloat(st[0] as f64),
        OrderedFloat(st[1] as f64),
        OrderedFloat(st[2] as f64),
        OrderedFloat(st[3] as f64),
    )
}

impl CartPoleEnv {
    /// `CartPoleEnv::new(render_mode)` (cartpole.rs:91-144) with the defaults of the paper.
    pub fn new(render_mode: RenderMode) -> Self {
        assert!(render_mode == RenderMode::None, "the GPU path renders nothing: use RenderMode::None");
        let mut p = std::mem::MaybeUninit::<CartPoleParams>::uninit();
        let status = unsafe { crate::ffi::gymrs_default_params(crate::ffi::GYMRS_CARTPOLE, p.as_mut_ptr() as *mut _) };
        assert_eq!(status, 0, "gymrs_default_params");
        let p = unsafe { p.assume_init() };
        let (rng, seed) = rand_random(None);
        let mut engine = Engine::new(1, 0, 0, &p, 0);
        engine.reset(Some(seed), None);
        let state = observation(&engine.state(0, 1));
        // bounds of cartpole.rs:105-113: twice the thresholds, unbounded velocities
        let high = CartPoleObservation::new(
            OrderedFloat(p.x_threshold * 2.0),
            OrderedFloat(f64::INFINITY),
            OrderedFloat(p.theta_threshold_radians * 2.0),
            OrderedFloat(f64::INFINITY),
        );
        CartPoleEnv {
            action_space: Discrete(2),
            observation_space: BoxR::new(-high, high),
            render_mode,
            state,
            metadata: Metadata::new(RENDER_MODES, 50), // cartpole.rs:265-269 (its Default impl is for the reference's own type)
            gravity: OrderedFloat(p.gravity),
            masscart: OrderedFloat(p.masscart),
            masspole: OrderedFloat(p.masspole),
            length: OrderedFloat(p.length),
            force_mag: OrderedFloat(p.force_mag),
            tau: OrderedFloat(p.tau),
            kinematics_integrator: KinematicsIntegrator::Euler,
            theta_threshold_radians: OrderedFloat(p.theta_threshold_radians),
            x_threshold: OrderedFloat(p.x_threshold),
            steps_beyond_terminated: None,
            pushed: p,
            engine,
            rand_random: rng,
        }
    }

    fn params_now(&self) -> CartPoleParams {
        CartPoleParams {
            gravity: self.gravity.into_inner(),
            masscart: self.masscart.into_inner(),
            masspole: self.masspole.into_inner(),
            length: self.length.into_inner(),
            force_mag: self.force_mag.into_inner(),
            tau: self.tau.into_inner(),
            theta_threshold_radians: self.theta_threshold_radians.into_inner(),
            x_threshold: self.x_threshold.into_inner(),
            kinematics_integrator: match self.kinematics_integrator {
                KinematicsIntegrator::Euler => 0,
                KinematicsIntegrator::Other => 1,
            },
            max_episode_steps: self.pushed.max_episode_steps,
        }
    }

    /// Push edited pub fields (constants or `state`) down to the device before the next step.
    fn sync_down(&mut self) {
        let now = self.params_now();
        if now != self.pushed {
            // only the launch constants change: the engine, its device, steps_beyond_terminated, seed and tick stay
            self.engine.set_params(&now);
            self.pushed = now;
        }
        let host: Vec<f64> = self.state.into();
        let st: Vec<f32> = host.iter().map(|v| *v as f32).collect();
        self.engine.set_state(0, 1, &st);
    }
}

impl Env for CartPoleEnv {
    type Action = usize;
    type Observation = CartPoleObservation;
    type Info = ();
    type ResetInfo = ();

    fn step(&mut self, action: usize) -> ActionReward<CartPoleObservation, ()> {
        assert!(self.action_space.contains(action), "{} (usize) invalid", action); // cartpole.rs:402-406
        self.sync_down();
        self.engine.step_host(&[action as u8]);
        self.state = observation(&self.engine.state(0, 1));
        let r = self.engine.lane_result(0);
        if r.done {
            // cartpole.rs:455-464: Some(0) on the terminating step, then counting
            self.steps_beyond_terminated = Some(self.steps_beyond_terminated.map_or(0, |k| k + 1));
        }
        ActionReward { observation: self.state, reward: OrderedFloat(r.reward as f64), done: r.done, truncated: false, info: Some(()) }
    }

    fn reset(
        &mut self,
        seed: Option<u64>,
        return_info: bool,
        options: Option<BoxR<CartPoleObservation>>,
    ) -> (CartPoleObservation, Option<()>) {
        let (rng, seed_no) = rand_random(seed);
        self.rand_random = rng;
        self.sync_down();
        let bounds: Option<Vec<f64>> = options.map(|b| {
            let (low, high): (Vec<f64>, Vec<f64>) = (b.low.into(), b.high.into());
            low.into_iter().chain(high).collect()
        });
        if cfg!(feature = "pcg64-reset") {
            // the state gym-rs itself returns for this seed (its Pcg64 + Uniform chain on the device), rounded to f32
            self.engine.reset_pcg64(Some(seed_no), bounds.as_deref());
            // ... and `rand_random()` hands out the generator where the reference's stands after its four draws (cartpole.rs:317-324)
            for _ in 0..4 {
                rand::RngCore::next_u64(&mut self.rand_random);
            }
        } else {
            let narrow: Option<Vec<f32>> = bounds.map(|b| b.iter().map(|v| *v as f32).collect());
            self.engine.reset(Some(seed_no), narrow.as_deref());
        }
        self.state = observation(&self.engine.state(0, 1));
        self.steps_beyond_terminated = None;
        (self.state, if return_info { Some(()) } else { None })
    }

    fn render(&mut self, _mode: RenderMode) -> Renders {
        Renders::None
    }

    fn close(&mut self) {}
}

impl EnvProperties for CartPoleEnv {
    type ActionSpace = Discrete;
    type ObservationSpace = BoxR<CartPoleObservation>;

    fn metadata(&self) -> &Metadata<Self> {
        &self.metadata
    }
    fn rand_random(&self) -> &Pcg64 {
        &self.rand_random
    }
    fn action_space(&self) -> &Discrete {
        &self.action_space
    }
    fn observation_space(&self) -> &BoxR<CartPoleObservation> {
        &self.observation_space
    }
    fn render_mode(&self) -> &RenderMode {
        &self.render_mode
    }
}
