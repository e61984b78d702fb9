#!/usr/bin/env python
"""Closed-loop use of the batched stepper from PyTorch: observation -> policy -> action -> step, all on one HIP
stream, no host synchronisation and no copies in between.

The observation columns are the engine's own SoA state arrays (for CartPole the observation IS the state,
cartpole.rs:476-482), wrapped zero-copy as torch tensors through the CUDA array interface.  The policy here is a
fixed linear controller (push the cart towards the side the pole falls to); any torch module that writes a uint8
action tensor works the same way.

    python examples/closed_loop_policy.py [--n-envs 1048576] [--steps 2000]
"""
from __future__ import annotations

import argparse
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")


class DeviceColumn:
    """A device array owned by the engine, presented to torch without a copy."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 3}


def column(ptr: int, n: int, typestr: str = "<f4") -> torch.Tensor:
    return torch.as_tensor(DeviceColumn(ptr, n, typestr), device="cuda:0")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=2000)
    args = ap.parse_args()
    n = args.n_envs

    stream = torch.cuda.Stream()
    env = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS)
    env.set_stream(stream.cuda_stream)  # the engine now launches on torch's stream
    env.reset(seed=0)
    x, x_dot, theta, theta_dot = (column(p, n) for p in env.obs_ptrs())
    reward = column(env.reward_ptr, n)
    done = column(env.done_ptr, n, "|u1")
    action = torch.empty(n, dtype=torch.uint8, device="cuda:0")

    with torch.cuda.stream(stream):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            # policy: a = 1 (push right) when the pole leans or moves to the right
            torch.gt(theta + 0.5 * theta_dot + 0.05 * x_dot + 0.01 * x, 0.0, out=action.view(torch.bool))
            env.step(action.data_ptr())
        env.sync()
        dt = time.perf_counter() - t0
    sum_return, sum_length, n_episodes, n_steps = env.stats()
    print(f"{n} envs x {args.steps} steps in {dt * 1e3:.1f} ms = {n_steps / dt:.3e} env-steps/s (policy included)")
    print(f"finished episodes: {int(n_episodes)}, mean return {sum_return / max(n_episodes, 1):.1f} "
          f"(a random policy gets about 22); last step: mean reward {reward.mean().item():.2f}, done {int(done.sum())}")
    env.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
