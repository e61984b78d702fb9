//! `MountainCarEnv` with the reference's public surface (mountain_car.rs:46-80, 392-530), one GPU lane behind it.
use crate::engine::Engine;
use crate::ffi::MountainCarParams;
use gym_rs::core::{ActionReward, Env, EnvProperties};
use gym_rs::envs::classical_control::mountain_car::MountainCarObservation;
use gym_rs::spaces::{BoxR, Discrete, Space};
use gym_rs::utils::custom::structs::Metadata;
use gym_rs::utils::custom::types::O64;
use gym_rs::utils::renderer::{RenderMode, Renders};
use gym_rs::utils::seeding::rand_random;
use ordered_float::OrderedFloat;
use rand_pcg::Pcg64;
use serde::Serialize;

/// GPU-backed drop-in for `gym_rs::envs::classical_control::mountain_car::MountainCarEnv`.
#[derive(Debug, Clone, Serialize)]
pub struct MountainCarEnv {
    /// Leftmost position.
    pub min_position: O64,
    /// Rightmost position.
    pub max_position: O64,
    /// Speed limit (both directions).
    pub max_speed: O64,
    /// Position of the flag.
    pub goal_position: O64,
    /// Velocity required at the flag.
    pub goal_velocity: O64,
    /// Force per action unit.
    pub force: O64,
    /// Gravity constant.
    pub gravity: O64,
    /// Always `RenderMode::None`.
    pub render_mode: RenderMode,
    /// `Discrete(3)`: push left, no push, push right.
    pub action_space: Discrete,
    /// Range of positions and velocities.
    pub observation_space: BoxR<MountainCarObservation>,
    /// The current state (host copy).
    pub state: MountainCarObservation,
    /// Additional pieces of information provided by the environment.
    pub metadata: Metadata<Self>,
    #[serde(skip_serializing)]
    pushed: MountainCarParams,
    #[serde(skip_serializing)]
    engine: Engine,
    #[serde(skip_serializing)]
    rand_random: Pcg64,
}

/// What the reference's metadata lists (mountain_car.rs:108-113); only `RenderMode::None` is honoured here.
const RENDER_MODES: &[RenderMode] = &[RenderMode::Human, RenderMode::RgbArray, RenderMode::SingleRgbArray, RenderMode::None];

fn observation(st: &[f32]) -> MountainCarObservation {
    MountainCarObservation::new(OrderedFloat(st[0] as f64), OrderedFloat(st[1] as f64))
}

impl MountainCarEnv {
    /// `MountainCarEnv::new(render_mode)` (mountain_car.rs:341-390).
    pub fn new(render_mode: RenderMode) -> Self {
        assert!(render_mode == RenderMode::None, "the GPU path renders nothing: use RenderMode::None");
        let mut p = std::mem::MaybeUninit::<MountainCarParams>::uninit();
        let status = unsafe { crate::ffi::gymrs_default_params(crate::ffi::GYMRS_MOUNTAIN_CAR, p.as_mut_ptr() as *mut _) };
        assert_eq!(status, 0, "gymrs_default_params");
        let p = unsafe { p.assume_init() };
        let (rng, seed) = rand_random(None);
        let mut engine = Engine::new(1, 0, 0, &p, 0);
        engine.reset(Some(seed), None);
        let state = observation(&engine.state(0, 1));
        MountainCarEnv {
            min_position: OrderedFloat(p.min_position),
            max_position: OrderedFloat(p.max_position),
            max_speed: OrderedFloat(p.max_speed),
            goal_position: OrderedFloat(p.goal_position),
            goal_velocity: OrderedFloat(p.goal_velocity),
            force: OrderedFloat(p.force),
            gravity: OrderedFloat(p.gravity),
            render_mode,
            action_space: Discrete(3),
            observation_space: BoxR::new(
                MountainCarObservation::new(OrderedFloat(p.min_position), OrderedFloat(-p.max_speed)),
                MountainCarObservation::new(OrderedFloat(p.max_position), OrderedFloat(p.max_speed)),
            ),
            state,
            metadata: Metadata::new(RENDER_MODES, 30), // mountain_car.rs:108-118 (its Default impl is for the reference's own type)
            pushed: p,
            engine,
            rand_random: rng,
        }
    }

    fn params_now(&self) -> MountainCarParams {
        MountainCarParams {
            min_position: self.min_position.into_inner(),
            max_position: self.max_position.into_inner(),
            max_speed: self.max_speed.into_inner(),
            goal_position: self.goal_position.into_inner(),
            goal_velocity: self.goal_velocity.into_inner(),
            force: self.force.into_inner(),
            gravity: self.gravity.into_inner(),
            max_episode_steps: self.pushed.max_episode_steps,
            _pad: 0,
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
        let st = [self.state.position.into_inner() as f32, self.state.velocity.into_inner() as f32];
        self.engine.set_state(0, 1, &st);
    }
}

impl Env for MountainCarEnv {
    type Action = usize;
    type Observation = MountainCarObservation;
    type Info = ();
    type ResetInfo = ();

    fn step(&mut self, action: usize) -> ActionReward<MountainCarObservation, ()> {
        assert!(self.action_space.contains(action), "{} (usize) invalid", action); // mountain_car.rs:402-406
        self.sync_down();
        self.engine.step_host(&[action as u8]);
        self.state = observation(&self.engine.state(0, 1));
        let r = self.engine.lane_result(0);
        ActionReward { observation: self.state, reward: OrderedFloat(r.reward as f64), done: r.done, truncated: false, info: None } // mountain_car.rs:433
    }

    fn reset(
        &mut self,
        seed: Option<u64>,
        return_info: bool,
        options: Option<BoxR<MountainCarObservation>>,
    ) -> (MountainCarObservation, Option<()>) {
        let (rng, seed_no) = rand_random(seed);
        self.rand_random = rng;
        self.sync_down();
        // only the position is sampled (mountain_car.rs:464-501); the velocity bounds are carried but unused
        let bounds: Option<[f64; 4]> = options.map(|b| {
            [b.low.position.into_inner(), b.low.velocity.into_inner(), b.high.position.into_inner(), b.high.velocity.into_inner()]
        });
        if cfg!(feature = "pcg64-reset") {
            // the position gym-rs itself draws for this seed (its Pcg64 + Uniform chain on the device), rounded to f32
            self.engine.reset_pcg64(Some(seed_no), bounds.as_ref().map(|b| &b[..]));
            // ... and `rand_random()` hands out the generator where the reference's stands after its one draw (mountain_car.rs:145)
            for _ in 0..1 {
                rand::RngCore::next_u64(&mut self.rand_random);
            }
        } else {
            let narrow: Option<[f32; 4]> = bounds.map(|b| [b[0] as f32, b[1] as f32, b[2] as f32, b[3] as f32]);
            self.engine.reset(Some(seed_no), narrow.as_ref().map(|b| &b[..]));
        }
        self.state = observation(&self.engine.state(0, 1));
        (self.state, if return_info { Some(()) } else { None })
    }

    fn render(&mut self, _mode: RenderMode) -> Renders {
        Renders::None
    }

    fn close(&mut self) {}
}

impl EnvProperties for MountainCarEnv {
    type ActionSpace = Discrete;
    type ObservationSpace = BoxR<MountainCarObservation>;

    fn metadata(&self) -> &Metadata<Self> {
        &self.metadata
    }
    fn rand_random(&self) -> &Pcg64 {
        &self.rand_random
    }
    fn action_space(&self) -> &Discrete {
        &self.action_space
    }
    fn observation_space(&self) -> &BoxR<MountainCarObservation> {
        &self.observation_space
    }
    fn render_mode(&self) -> &RenderMode {
        &self.render_mode
    }
}
