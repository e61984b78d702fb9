"""One-off long-run check of the CHAIN path (a script next to the suite, not collected by pytest): two engines of 2^20 lanes (CartPole with all three
flags, then MountainCar, then Pendulum), same seed, same action ring -- one stepped through chains of the engine's own dispatcher in calls of random
length (8 .. 20 000 launches), the other through HIP launches (GYMRS_AQL is looked up per call).  Every chain launch runs the per-launch XCD check;
after every round (~2e5 steps) state bits, step results and statistics of the two must be equal, and no chain may have reported an error.
    gpurun -- 'python tests/soak_chains.py [steps per env = 2000000] [lanes_log2 = 20]'   (from 22 on CartPole elides its reward store)
"""
import importlib
import json
import os
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

gymrs = importlib.import_module("gym-rs_amd")
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n, nbuf = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20), 8
rng = random.Random(7)
for kind, name in ((0, "cartpole"), (1, "mountain_car"), (2, "pendulum")):
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | gymrs.TIME_LIMIT
    esz = 4 if kind == 2 else 1
    a = gymrs.BatchedEngine(kind, n, flags=flags)
    b = gymrs.BatchedEngine(kind, n, flags=flags)
    ring = torch.empty(nbuf * n * esz, dtype=torch.uint8, device="cuda:0")
    for k in range(nbuf):
        a.fill_actions(ring.data_ptr() + k * n * esz, seed=3, t=k)
    a.reset(seed=11)
    b.reset(seed=11)
    t0 = time.time()
    done, rounds, calls = 0, 0, 0
    while done < total:
        round_steps = 0
        while round_steps < min(200_000, total // 4):
            k = rng.choice((8, 9, 31, 100, 1000, 4999, 20000))
            os.environ["GYMRS_AQL"] = "1"
            a.step_many(ring.data_ptr(), n * esz, nbuf, k)
            os.environ["GYMRS_AQL"] = "0"
            b.step_many(ring.data_ptr(), n * esz, nbuf, k)
            round_steps += k
            calls += 1
        a.sync()
        b.sync()
        done += round_steps
        rounds += 1
        assert np.array_equal(a.get_state().view(np.uint32), b.get_state().view(np.uint32)), (name, done)
        ra, rb = a.get_step_result(), b.get_step_result()
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)), (name, done)
        assert np.array_equal(a.stats(), b.stats()), (name, done, a.stats(), b.stats())
    os.environ["GYMRS_AQL"] = "1"
    ex = json.loads(a.env_json(0))["gymrs"]
    exb = json.loads(b.env_json(0))["gymrs"]
    assert ex["aql_launches"] >= done * 0.99 and exb["aql_launches"] == 0, (ex, exb)
    print(f"chain soak ok: {name}: {done} steps of {n} lanes in {calls} calls, {ex['aql_chains']} chains / {ex['aql_launches']} chain launches (each with the XCD check) "
          f"== HIP launches after each of {rounds} rounds; dispatcher: {ex['aql']}, hand-over {ex['aql_handover']}; {time.time() - t0:.0f} s", flush=True)
    a.close()
    b.close()
