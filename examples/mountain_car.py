#!/usr/bin/env python
"""The caller loop of the reference's examples/mountain_car.rs:8-40 through the single-env mirror: random actions until
the episode is done or 200 steps have passed, close, then -- like the reference -- 200 more steps on the closed env
(Env::close only tears the GUI down; stepping stays legal).

    python examples/mountain_car.py        # needs an MI355X; RenderMode.NONE only
"""
import importlib
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402

gymrs = importlib.import_module("gym-rs_amd")


def main(seed: int = 0, verbose: bool = True):
    mc = gymrs.MountainCarEnv(gymrs.RenderMode.NONE)
    _state = mc.reset(None, False, None)
    rng = random.Random(seed)
    end = False
    episode_length = 0
    while not end:
        if episode_length > 200:
            break
        action = rng.randrange(3)  # rng.gen_range(0..3)
        ar = mc.step(action)
        episode_length += 1
        end = ar.done
        if verbose:
            print("episode_length:", episode_length)
    mc.close()  # mountain_car.rs:503-505: drops the GUI handle only
    for _ in range(200):  # examples/mountain_car.rs:30-37: the env is still steppable
        action = rng.randrange(3)
        mc.step(action)
        episode_length += 1
        if verbose:
            print("episode_length:", episode_length)
    mc.release()
    return episode_length


if __name__ == "__main__":
    print("total steps:", main(verbose=False))
