//! One batch over several GPUs in ONE process: safe RAII wrapper of `gymrs_sharded` (include/gymrs_amd.h, ABI 3).
//!
//! gym-rs' envs are independent of each other (no cross-env term anywhere in cartpole.rs:398-483 / mountain_car.rs:398-435), so a batch of
//! `n_total` lanes is cut into contiguous blocks -- one engine and one native host thread per block, bound to the block's device -- with the lane's
//! GLOBAL id in the Philox counters: every result is bit-identical to one [`crate::engine::Engine`] of `n_total` lanes.  The only exchange is the four
//! statistics doubles: a grouped RCCL all-reduce over xGMI when the blocks sit on distinct devices, a host-side sum where they share one.
//! NOT COMPILED in the authoring image (no Rust toolchain); it follows `engine.rs`, which is reviewed against `core.rs:25-90` the same way.
use crate::engine::Kind;
use crate::ffi;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};

fn check(status: c_int) {
    if status != ffi::GYMRS_OK {
        let msg = unsafe { CStr::from_ptr(ffi::gymrs_last_error()) }.to_string_lossy().into_owned();
        panic!("gymrs_amd: status {status}: {msg}"); // the reference has no Results: every failure is a panic (cartpole.rs:402-406)
    }
}

/// One block of a [`ShardedEngine`]: where its lanes sit in the batch and on which GPU.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Block {
    /// first lane of the block within the batch
    pub first_lane: u64,
    /// lanes in the block
    pub n_lanes: u64,
    /// HIP device index
    pub device: i32,
}

/// Owns the sharder (its engines and host threads); dropping it releases everything (`Env::close`, core.rs:56).
/// `&mut self` everywhere: one caller thread at a time, like the reference's `Env` methods (core.rs:42-50).
#[derive(Debug)]
pub struct ShardedEngine {
    raw: *mut ffi::GymrsSharded,
    kind: Kind,
    n_total: u64,
    blocks: Vec<Block>,
}

impl ShardedEngine {
    /// `gymrs_sharded_create` with the kind's default constants: block `r` lives on `devices[r]` (a device may repeat).
    pub fn with_defaults(kind: Kind, n_total: u64, global_env_offset: u64, devices: &[i32], flags: u32) -> Self {
        let raw_kind = match kind {
            Kind::CartPole => ffi::GYMRS_CARTPOLE,
            Kind::MountainCar => ffi::GYMRS_MOUNTAIN_CAR,
        };
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::gymrs_sharded_create(raw_kind, n_total, global_env_offset, devices.len() as c_int, devices.as_ptr(), std::ptr::null(), flags, &mut raw)
        });
        let mut blocks = Vec::with_capacity(devices.len());
        for r in 0..devices.len() {
            let (mut first, mut count, mut dev) = (0u64, 0u64, 0 as c_int);
            check(unsafe { ffi::gymrs_sharded_shard(raw, r as c_int, std::ptr::null_mut(), &mut first, &mut count, &mut dev) });
            blocks.push(Block { first_lane: first, n_lanes: count, device: dev });
        }
        ShardedEngine { raw, kind, n_total, blocks }
    }

    /// The blocks, in lane order.
    pub fn blocks(&self) -> &[Block] {
        &self.blocks
    }

    /// Lanes of the whole batch.
    pub fn len(&self) -> u64 {
        self.n_total
    }

    /// `true` for a batch without lanes (cannot be created; for clippy's `len_without_is_empty`).
    pub fn is_empty(&self) -> bool {
        self.n_total == 0
    }

    /// `Env::reset` for the whole batch with ONE seed; returns the seed used (seeding.rs:21-26).
    pub fn reset(&mut self, seed: Option<u64>) -> u64 {
        let mut used = 0u64;
        check(unsafe { ffi::gymrs_sharded_reset(self.raw, seed.is_some() as c_int, seed.unwrap_or(0), std::ptr::null(), &mut used) });
        used
    }

    /// One asynchronous `Env::step` of every lane; `actions_dev[r]` = block r's action buffer on ITS device.
    ///
    /// # Safety
    /// Every pointer must be a device address (on the block's device) of at least the block's `n_lanes` bytes, valid until `sync()`.
    pub unsafe fn step_device(&mut self, actions_dev: &[*const c_void]) {
        assert_eq!(actions_dev.len(), self.blocks.len());
        check(ffi::gymrs_sharded_step(self.raw, actions_dev.as_ptr()));
    }

    /// `n_steps` consecutive steps, step `t` taking block r's actions from `actions_dev[r] + (t % n_buffers) * stride_bytes`.
    ///
    /// # Safety
    /// As [`Self::step_device`], for `n_buffers` buffers per block.
    pub unsafe fn step_many(&mut self, actions_dev: &[*const c_void], stride_bytes: u64, n_buffers: u32, n_steps: u32) {
        assert_eq!(actions_dev.len(), self.blocks.len());
        check(ffi::gymrs_sharded_step_many(self.raw, actions_dev.as_ptr(), stride_bytes, n_buffers, n_steps, 0));
    }

    /// Random-policy actions (`rng.gen_range(0..=1)`, examples/cartpole.rs:19) for time `t` into every block's buffer.
    ///
    /// # Safety
    /// As [`Self::step_device`], writable.
    pub unsafe fn fill_actions(&mut self, actions_dev: &[*mut c_void], seed: u64, t: u64) {
        assert_eq!(actions_dev.len(), self.blocks.len());
        check(ffi::gymrs_sharded_fill_actions(self.raw, actions_dev.as_ptr(), seed, t));
    }

    /// `n_steps` random-policy steps of every lane fused into one launch per block (the loop of examples/cartpole.rs:15-30 per lane).
    pub fn rollout(&mut self, n_steps: u32, action_seed: u64, action_t0: u64) {
        check(unsafe { ffi::gymrs_sharded_rollout(self.raw, n_steps, action_seed, action_t0) });
    }

    /// Wait for every block's stream; panics like the reference's `assert!` if a step saw an action outside the action space.
    pub fn sync(&mut self) {
        check(unsafe { ffi::gymrs_sharded_sync(self.raw) });
    }

    /// `[sum_return, sum_length, n_episodes, n_steps]` of the whole batch.
    pub fn stats(&mut self) -> [f64; 4] {
        let mut out = [0f64; 4];
        check(unsafe { ffi::gymrs_sharded_stats(self.raw, out.as_mut_ptr()) });
        out
    }

    /// How the last `stats()` summed: `"rccl"`, `"host"` or `"none"`.
    pub fn reduce_path(&mut self) -> String {
        unsafe { CStr::from_ptr(ffi::gymrs_sharded_reduce_path(self.raw)) }.to_string_lossy().into_owned()
    }

    /// State of lanes `[first, first + count)` of the batch, SoA (`state_dim` arrays of `count` floats).
    pub fn state(&mut self, first: u64, count: u64) -> Vec<f32> {
        let mut out = vec![0f32; count as usize * self.kind.state_dim()];
        check(unsafe { ffi::gymrs_sharded_get_state(self.raw, first, count, out.as_mut_ptr()) });
        out
    }
}

impl Drop for ShardedEngine {
    fn drop(&mut self) {
        unsafe { ffi::gymrs_sharded_destroy(self.raw) };
    }
}
