//! Raw declarations of `include/gymrs_amd.h` (ABI version 3).  Field order and types follow the header.
#![allow(missing_docs)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct GymrsEngine {
    _opaque: [u8; 0],
}

/// `gymrs_sharded`: one batch over several GPUs in one process.
#[repr(C)]
pub struct GymrsSharded {
    _opaque: [u8; 0],
}

pub const GYMRS_OK: c_int = 0;
pub const GYMRS_EACTION: c_int = 5;

pub const GYMRS_CARTPOLE: c_int = 0;
pub const GYMRS_MOUNTAIN_CAR: c_int = 1;
pub const GYMRS_PENDULUM: c_int = 2;

pub const GYMRS_AUTO_RESET: u32 = 1;
pub const GYMRS_TRACK_STATS: u32 = 2;
pub const GYMRS_TIME_LIMIT: u32 = 4;

/// `gymrs_cartpole_params`: the pub physics fields of `CartPoleEnv` (cartpole.rs:53-82), f64 like the reference.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct CartPoleParams {
    pub gravity: f64,
    pub masscart: f64,
    pub masspole: f64,
    pub length: f64,
    pub force_mag: f64,
    pub tau: f64,
    pub theta_threshold_radians: f64,
    pub x_threshold: f64,
    /// 0 = Euler, 1 = semi-implicit (`KinematicsIntegrator`, cartpole.rs:380-387)
    pub kinematics_integrator: i32,
    pub max_episode_steps: u32,
}

/// `gymrs_mountain_car_params`: the pub physics fields of `MountainCarEnv` (mountain_car.rs:49-77).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct MountainCarParams {
    pub min_position: f64,
    pub max_position: f64,
    pub max_speed: f64,
    pub goal_position: f64,
    pub goal_velocity: f64,
    pub force: f64,
    pub gravity: f64,
    pub max_episode_steps: u32,
    pub _pad: u32,
}

/// `gymrs_trajectory`: device buffers `[n_steps][..][lane_stride]` filled by `gymrs_rollout_record`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Trajectory {
    pub obs: *mut f32,
    pub actions: *mut c_void,
    pub reward: *mut f32,
    pub done: *mut u8,
    pub truncated: *mut u8,
    pub lane_stride: u64,
}

extern "C" {
    pub fn gymrs_abi_version() -> c_int;
    pub fn gymrs_last_error() -> *const c_char;
    pub fn gymrs_default_params(kind: c_int, params: *mut c_void) -> c_int;
    pub fn gymrs_observation_space(kind: c_int, params: *const c_void, low: *mut f64, high: *mut f64, dim: *mut c_int) -> c_int;
    pub fn gymrs_engine_create(
        kind: c_int,
        n_envs: u64,
        global_env_offset: u64,
        device: c_int,
        params: *const c_void,
        flags: u32,
        out: *mut *mut GymrsEngine,
    ) -> c_int;
    pub fn gymrs_engine_destroy(e: *mut GymrsEngine) -> c_int;
    pub fn gymrs_engine_clone(src: *mut GymrsEngine, out: *mut *mut GymrsEngine) -> c_int;
    pub fn gymrs_reset(e: *mut GymrsEngine, has_seed: c_int, seed: u64, bounds_low_high: *const f32, seed_used: *mut u64) -> c_int;
    pub fn gymrs_reset_pcg64(
        e: *mut GymrsEngine,
        has_seed: c_int,
        seed: u64,
        seeds_dev: *const u64,
        bounds_low_high: *const f64,
        seed_used: *mut u64,
    ) -> c_int;
    pub fn gymrs_step(e: *mut GymrsEngine, actions_dev: *const c_void) -> c_int;
    pub fn gymrs_step_host(e: *mut GymrsEngine, actions_host: *const c_void) -> c_int;
    pub fn gymrs_step_many(
        e: *mut GymrsEngine,
        actions_dev: *const c_void,
        stride_bytes: u64,
        n_buffers: u32,
        n_steps: u32,
        use_graph: c_int,
    ) -> c_int;
    pub fn gymrs_rollout(e: *mut GymrsEngine, n_steps: u32, action_seed: u64, action_t0: u64) -> c_int;
    pub fn gymrs_rollout_record(e: *mut GymrsEngine, n_steps: u32, action_seed: u64, action_t0: u64, out: *const Trajectory) -> c_int;
    pub fn gymrs_sync(e: *mut GymrsEngine) -> c_int;
    pub fn gymrs_get_state(e: *mut GymrsEngine, first: u64, count: u64, host_out: *mut f32) -> c_int;
    pub fn gymrs_set_state(e: *mut GymrsEngine, first: u64, count: u64, host_in: *const f32) -> c_int;
    pub fn gymrs_get_step_result(
        e: *mut GymrsEngine,
        first: u64,
        count: u64,
        reward: *mut f32,
        done: *mut u8,
        truncated: *mut u8,
    ) -> c_int;
    pub fn gymrs_stats(e: *mut GymrsEngine, out4: *mut f64) -> c_int;
    pub fn gymrs_snapshot_size(e: *mut GymrsEngine, bytes: *mut u64) -> c_int;
    pub fn gymrs_snapshot_save(e: *mut GymrsEngine, host_buf: *mut c_void, bytes: u64) -> c_int;
    pub fn gymrs_snapshot_load(e: *mut GymrsEngine, host_buf: *const c_void, bytes: u64) -> c_int;
    // ABI 2
    pub fn gymrs_set_params(e: *mut GymrsEngine, params: *const c_void) -> c_int;
    pub fn gymrs_get_params(e: *mut GymrsEngine, params_out: *mut c_void) -> c_int;
    pub fn gymrs_env_json(e: *mut GymrsEngine, lane: u64, buf: *mut c_char, cap: u64, needed: *mut u64) -> c_int;
    pub fn gymrs_params_from_json(kind: c_int, json: *const c_char, params: *mut c_void, state: *mut f64, state_dim: *mut c_int) -> c_int;
    // ABI 3: one batch over several GPUs in ONE process (one engine + one native host thread per block)
    pub fn gymrs_allreduce_stats_multi(shards: *mut *mut GymrsEngine, n: c_int, out4: *mut f64, used_rccl: *mut c_int) -> c_int;
    pub fn gymrs_sharded_create(
        kind: c_int,
        n_total: u64,
        global_env_offset: u64,
        n_shards: c_int,
        devices: *const c_int,
        params: *const c_void,
        flags: u32,
        out: *mut *mut GymrsSharded,
    ) -> c_int;
    pub fn gymrs_sharded_destroy(h: *mut GymrsSharded) -> c_int;
    pub fn gymrs_sharded_count(h: *mut GymrsSharded, n_shards: *mut c_int) -> c_int;
    pub fn gymrs_sharded_shard(
        h: *mut GymrsSharded,
        shard: c_int,
        engine: *mut *mut GymrsEngine,
        first_lane: *mut u64,
        n_lanes: *mut u64,
        device: *mut c_int,
    ) -> c_int;
    pub fn gymrs_sharded_reset(h: *mut GymrsSharded, has_seed: c_int, seed: u64, bounds_low_high: *const f32, seed_used: *mut u64) -> c_int;
    pub fn gymrs_sharded_step(h: *mut GymrsSharded, actions_dev: *const *const c_void) -> c_int;
    pub fn gymrs_sharded_step_many(
        h: *mut GymrsSharded,
        actions_dev: *const *const c_void,
        stride_bytes: u64,
        n_buffers: u32,
        n_steps: u32,
        use_graph: c_int,
    ) -> c_int;
    pub fn gymrs_sharded_fill_actions(h: *mut GymrsSharded, actions_dev: *const *mut c_void, seed: u64, t: u64) -> c_int;
    pub fn gymrs_sharded_rollout(h: *mut GymrsSharded, n_steps: u32, action_seed: u64, action_t0: u64) -> c_int;
    pub fn gymrs_sharded_set_params(h: *mut GymrsSharded, params: *const c_void) -> c_int;
    pub fn gymrs_sharded_sync(h: *mut GymrsSharded) -> c_int;
    pub fn gymrs_sharded_stats(h: *mut GymrsSharded, out4: *mut f64) -> c_int;
    pub fn gymrs_sharded_stats_clear(h: *mut GymrsSharded) -> c_int;
    pub fn gymrs_sharded_reduce_path(h: *mut GymrsSharded) -> *const c_char;
    pub fn gymrs_sharded_get_state(h: *mut GymrsSharded, first: u64, count: u64, host_out: *mut f32) -> c_int;
    pub fn gymrs_sharded_get_step_result(h: *mut GymrsSharded, first: u64, count: u64, reward: *mut f32, done: *mut u8, truncated: *mut u8) -> c_int;
}
