#!/usr/bin/env python
"""The caller loop of the reference's examples/cartpole.rs:7-33, first through the single-env mirror (same calls, one GPU
lane: plumbing, BASELINE.json configs[0]), then for 2^20 envs at once: the same loop -- random action, step, reset on
done, sum the rewards of an episode -- as ONE fused launch per 475 steps (gymrs_rollout) with the episode statistics read
back at the end.

    python examples/cartpole.py            # needs an MI355X; RenderMode.NONE only (the GUI is out of scope)
"""
import importlib
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402  (first, so the extension shares torch's HIP runtime)

gymrs = importlib.import_module("gym-rs_amd")


def single_env(n_episodes: int = 15, seed: int = 0):
    env = gymrs.CartPoleEnv(gymrs.RenderMode.NONE)  # examples/cartpole.rs:8 uses RenderMode::Human
    env.reset(None, False, None)
    rng = random.Random(seed)
    rewards = []
    for _ in range(n_episodes):
        current_reward = 0.0
        for _ in range(475):
            action = rng.randint(0, 1)  # rng.gen_range(0..=1)
            state_reward = env.step(action)
            current_reward += state_reward.reward
            if state_reward.done:
                break
        env.reset(None, False, None)
        rewards.append(current_reward)
    env.close()
    return rewards


def batched(n_envs: int = 1 << 20, steps: int = 475):
    eng = gymrs.BatchedEngine(gymrs.CARTPOLE, n_envs, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS)
    eng.reset(seed=0)
    eng.rollout(steps, action_seed=1)  # gen_range -> step -> reset on done -> reward sum, for every lane, in one launch
    sum_return, sum_length, n_episodes, n_steps = eng.stats()
    eng.close()
    return {"envs": n_envs, "steps": int(n_steps), "finished_episodes": int(n_episodes), "mean_return": float(sum_return / max(n_episodes, 1))}


if __name__ == "__main__":
    print(single_env())
    print(batched())
