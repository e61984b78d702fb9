//! Safe RAII wrapper of one `gymrs_engine` (one shard of lanes on one GPU, driven by one thread: `&mut self`).
use crate::ffi;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};

/// Which env the lanes of an engine are.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Kind {
    /// `CartPoleEnv` (cartpole.rs)
    CartPole,
    /// `MountainCarEnv` (mountain_car.rs)
    MountainCar,
}

impl Kind {
    fn raw(self) -> c_int {
        match self {
            Kind::CartPole => ffi::GYMRS_CARTPOLE,
            Kind::MountainCar => ffi::GYMRS_MOUNTAIN_CAR,
        }
    }
    /// Number of f32 state components per lane.
    pub fn state_dim(self) -> usize {
        match self {
            Kind::CartPole => 4,
            Kind::MountainCar => 2,
        }
    }
}

mod sealed {
    pub trait Sealed {}
    impl Sealed for crate::ffi::CartPoleParams {}
    impl Sealed for crate::ffi::MountainCarParams {}
}

/// The params struct of one env kind.  Sealed: only the two `#[repr(C)]` structs of [`crate::ffi`] implement it, each tied
/// to its [`Kind`], so a safe caller cannot hand `gymrs_engine_create` a struct of the wrong layout.
pub trait Params: sealed::Sealed + Copy {
    /// The env kind these params belong to.
    const KIND: Kind;
}
impl Params for ffi::CartPoleParams {
    const KIND: Kind = Kind::CartPole;
}
impl Params for ffi::MountainCarParams {
    const KIND: Kind = Kind::MountainCar;
}

/// The reference has no `Result`s: every failure is a panic (cartpole.rs:402-406, screen.rs:184-203).
/// So is every non-zero status of the C ABI here.
fn check(status: c_int) {
    if status != ffi::GYMRS_OK {
        let msg = unsafe { CStr::from_ptr(ffi::gymrs_last_error()) }.to_string_lossy().into_owned();
        panic!("gymrs_amd: status {status}: {msg}");
    }
}

/// Result of one step of one lane, as host values.
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct LaneResult {
    /// reward of the step
    pub reward: f32,
    /// the episode terminated
    pub done: bool,
    /// the episode hit the time limit (only with `GYMRS_TIME_LIMIT`)
    pub truncated: bool,
}

/// Owns the engine handle; dropping it releases the device memory (`Env::close`, core.rs:56).
#[derive(Debug)]
pub struct Engine {
    raw: *mut ffi::GymrsEngine,
    kind: Kind,
    n: u64,
    device: i32,
}

impl Engine {
    /// `gymrs_engine_create` with the kind's default constants.
    pub fn with_defaults(kind: Kind, n_envs: u64, global_env_offset: u64, device: i32, flags: u32) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::gymrs_engine_create(kind.raw(), n_envs, global_env_offset, device, std::ptr::null(), flags, &mut raw) });
        Engine { raw, kind, n: n_envs, device }
    }

    /// `gymrs_engine_create`; the env kind follows from the params type.
    pub fn new<P: Params>(n_envs: u64, global_env_offset: u64, device: i32, params: &P, flags: u32) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::gymrs_engine_create(P::KIND.raw(), n_envs, global_env_offset, device, params as *const P as *const c_void, flags, &mut raw)
        });
        Engine { raw, kind: P::KIND, n: n_envs, device }
    }

    /// Assign the pub physics fields of every lane (`gymrs_set_params`): only the constants change -- state,
    /// `steps_beyond_terminated`, the episode clock, seed and tick carry on, exactly like `env.gravity = ..` on the
    /// reference struct between two `step()` calls (cartpole.rs:455-464 reads the fields afresh on every step).
    pub fn set_params<P: Params>(&mut self, params: &P) {
        assert_eq!(P::KIND, self.kind, "params of another env kind");
        check(unsafe { ffi::gymrs_set_params(self.raw, params as *const P as *const c_void) });
    }

    /// The GPU this engine lives on.
    pub fn device(&self) -> i32 {
        self.device
    }

    /// What `serde_json::to_string(&env)` prints for the reference env that lane `lane` stands for (`gymrs_env_json`).
    pub fn env_json(&mut self, lane: u64) -> String {
        let mut need = 0u64;
        let mut buf = vec![0u8; 2048];
        let mut st = unsafe { ffi::gymrs_env_json(self.raw, lane, buf.as_mut_ptr() as *mut _, buf.len() as u64, &mut need) };
        if st != ffi::GYMRS_OK && need as usize > buf.len() {
            buf = vec![0u8; need as usize];
            st = unsafe { ffi::gymrs_env_json(self.raw, lane, buf.as_mut_ptr() as *mut _, buf.len() as u64, &mut need) };
        }
        check(st);
        let end = buf.iter().position(|b| *b == 0).unwrap_or(buf.len());
        String::from_utf8_lossy(&buf[..end]).into_owned()
    }

    /// Number of lanes.
    pub fn len(&self) -> u64 {
        self.n
    }

    /// `true` for an engine without lanes.
    pub fn is_empty(&self) -> bool {
        self.n == 0
    }

    /// `Env::reset` for every lane; returns the seed used (seeding.rs:21-26 echoes it the same way).
    pub fn reset(&mut self, seed: Option<u64>, bounds_low_high: Option<&[f32]>) -> u64 {
        if let Some(b) = bounds_low_high {
            assert_eq!(b.len(), 2 * self.kind.state_dim(), "bounds = state_dim lows then state_dim highs");
        }
        let mut used = 0u64;
        check(unsafe {
            ffi::gymrs_reset(
                self.raw,
                seed.is_some() as c_int,
                seed.unwrap_or(0),
                bounds_low_high.map_or(std::ptr::null(), |b| b.as_ptr()),
                &mut used,
            )
        });
        used
    }

    /// The same reset drawn from the reference's own generator chain (`Pcg64::seed_from_u64` + `Uniform` over f64,
    /// seeding.rs:21-26): lane i holds, rounded once to f32, what gym-rs' `reset(Some(seed + offset + i))` returns.
    /// CartPole / MountainCar only.
    pub fn reset_pcg64(&mut self, seed: Option<u64>, bounds_low_high: Option<&[f64]>) -> u64 {
        if let Some(b) = bounds_low_high {
            assert_eq!(b.len(), 2 * self.kind.state_dim(), "bounds = state_dim lows then state_dim highs");
        }
        let mut used = 0u64;
        check(unsafe {
            ffi::gymrs_reset_pcg64(
                self.raw,
                seed.is_some() as c_int,
                seed.unwrap_or(0),
                std::ptr::null(),
                bounds_low_high.map_or(std::ptr::null(), |b| b.as_ptr()),
                &mut used,
            )
        });
        used
    }

    /// One `Env::step` of every lane with host actions (one u8 per lane), then wait for it.
    pub fn step_host(&mut self, actions: &[u8]) {
        assert_eq!(actions.len() as u64, self.n);
        check(unsafe { ffi::gymrs_step_host(self.raw, actions.as_ptr() as *const c_void) });
        check(unsafe { ffi::gymrs_sync(self.raw) });
    }

    /// One asynchronous `Env::step` with a device-resident action buffer (one u8 per lane).
    ///
    /// # Safety
    /// `actions_dev` must be a device address of at least `len()` bytes that stays valid until `sync()`.
    pub unsafe fn step_device(&mut self, actions_dev: *const c_void) {
        check(ffi::gymrs_step(self.raw, actions_dev));
    }

    /// `n_steps` consecutive `Env::step`s, step `t` taking its actions from buffer `t % n_buffers` of a ring of device buffers
    /// `stride_bytes` apart: ONE call, asynchronous on the engine's stream.  HIP launches by default (since round 6); with `GYMRS_AQL=1` in the
    /// process environment calls of 8 steps and more are submitted as a chain through the engine's own HSA queue (release fence only at the
    /// end of the chain: 4.9 instead of 6.4 us per 2^20-lane CartPole step; one process per GPU -- INTEGRATION.md "gymrs_step_many").  A caller
    /// that has K action buffers ready should prefer this to K `step_device` calls either way (one FFI call, one host loop).
    ///
    /// # Safety
    /// `actions_dev` must be a device address of `n_buffers` buffers of at least `len()` bytes each, valid until `sync()`.
    pub unsafe fn step_many(&mut self, actions_dev: *const c_void, stride_bytes: u64, n_buffers: u32, n_steps: u32, use_graph: bool) {
        check(ffi::gymrs_step_many(self.raw, actions_dev, stride_bytes, n_buffers, n_steps, use_graph as c_int));
    }

    /// `n_steps` random-policy steps fused into one launch (the loop of examples/cartpole.rs:15-30 per lane).
    pub fn rollout(&mut self, n_steps: u32, action_seed: u64, action_t0: u64) {
        check(unsafe { ffi::gymrs_rollout(self.raw, n_steps, action_seed, action_t0) });
    }

    /// `rollout` that also keeps every step's observation / action / reward / done in device buffers.
    ///
    /// # Safety
    /// The pointers in `out` must be device buffers of the sizes `include/gymrs_amd.h` documents for `gymrs_trajectory`
    /// and stay valid until `sync()`.
    pub unsafe fn rollout_record(&mut self, n_steps: u32, action_seed: u64, action_t0: u64, out: &ffi::Trajectory) {
        check(ffi::gymrs_rollout_record(self.raw, n_steps, action_seed, action_t0, out));
    }

    /// Wait for everything queued on the engine's stream.
    pub fn sync(&mut self) {
        check(unsafe { ffi::gymrs_sync(self.raw) });
    }

    /// State of lanes `first..first+count`, component-major (`[component][lane]`).
    pub fn state(&mut self, first: u64, count: u64) -> Vec<f32> {
        let mut out = vec![0f32; self.kind.state_dim() * count as usize];
        check(unsafe { ffi::gymrs_get_state(self.raw, first, count, out.as_mut_ptr()) });
        out
    }

    /// Overwrite the state of lanes `first..first+count` (like assigning the pub `state` field).
    pub fn set_state(&mut self, first: u64, count: u64, component_major: &[f32]) {
        assert_eq!(component_major.len(), self.kind.state_dim() * count as usize);
        check(unsafe { ffi::gymrs_set_state(self.raw, first, count, component_major.as_ptr()) });
    }

    /// Reward / done / truncated of the last step of one lane.
    pub fn lane_result(&mut self, lane: u64) -> LaneResult {
        let (mut reward, mut done, mut truncated) = (0f32, 0u8, 0u8);
        check(unsafe { ffi::gymrs_get_step_result(self.raw, lane, 1, &mut reward, &mut done, &mut truncated) });
        LaneResult { reward, done: done != 0, truncated: truncated != 0 }
    }

    /// `[sum_return, sum_length, n_episodes, n_steps]` since the last reset / clear.
    pub fn stats(&mut self) -> [f64; 4] {
        let mut out = [0f64; 4];
        check(unsafe { ffi::gymrs_stats(self.raw, out.as_mut_ptr()) });
        out
    }

    /// Opaque blob holding everything a step can observe (`Serialize`).
    pub fn snapshot(&mut self) -> Vec<u8> {
        let mut bytes = 0u64;
        check(unsafe { ffi::gymrs_snapshot_size(self.raw, &mut bytes) });
        let mut buf = vec![0u8; bytes as usize];
        check(unsafe { ffi::gymrs_snapshot_save(self.raw, buf.as_mut_ptr() as *mut c_void, bytes) });
        buf
    }

    /// Load a `snapshot()` of an engine with the same kind, lane count and flags.
    pub fn restore(&mut self, blob: &[u8]) {
        check(unsafe { ffi::gymrs_snapshot_load(self.raw, blob.as_ptr() as *const c_void, blob.len() as u64) });
    }
}

impl Clone for Engine {
    /// Deep copy on the device (`gymrs_engine_clone`): state, episode bookkeeping, RNG position.
    fn clone(&self) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::gymrs_engine_clone(self.raw, &mut raw) });
        Engine { raw, kind: self.kind, n: self.n, device: self.device }
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        if !self.raw.is_null() {
            unsafe { ffi::gymrs_engine_destroy(self.raw) };
            self.raw = std::ptr::null_mut();
        }
    }
}
