//! MI355X-backed drop-ins for gym-rs' classic-control envs.
//!
//! * [`cartpole::CartPoleEnv`], [`mountain_car::MountainCarEnv`]: the reference's struct names, pub fields and
//!   `Env` / `EnvProperties` impls (gym-rs `src/core.rs`), one GPU lane each -- source compatibility for
//!   `examples/*.rs`; no faster than the CPU crate (a step is a kernel launch plus two small copies).
//! * [`engine::Engine`]: the batched stepper itself -- millions of lanes per GPU, `step_device`, `rollout`,
//!   `stats`, `clone`, `snapshot` -- which is what the C ABI is built for.
//! * [`sharded::ShardedEngine`]: one batch over the GPUs of a node in one process (one engine and one native host thread per block).
//!
//! Deviation: `EnvProperties::rand_random` returns a `Pcg64` that only mirrors the seed; the device samples
//! with counter-based Philox4x32-10 (state = seed + tick).
#![warn(missing_docs)]
pub mod cartpole;
pub mod engine;
pub mod ffi;
pub mod mountain_car;
pub mod sharded;
