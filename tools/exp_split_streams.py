"""Experiment: does partitioning the lanes over k engines on k streams (each replaying its own captured graph) overlap the
dependent-launch boundaries of the per-step kernel?  k = 1 is the product's shape.  Prints us per full-batch step."""
import argparse
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")


def run(kind, n_total, k, steps, nbuf, reps):
    n = n_total // k
    engs, bufs = [], []
    for i in range(k):
        e = gymrs.BatchedEngine(kind, n, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS, global_env_offset=i * n)
        e.reset(seed=1)
        b = torch.empty((nbuf, n), dtype=torch.uint8 if kind != 2 else torch.float32, device="cuda:0")
        for j in range(nbuf):
            e.fill_actions(b[j].data_ptr(), seed=2, t=j)
        engs.append(e)
        bufs.append(b)
    esz = bufs[0].element_size()
    best = []
    for r in range(reps + 1):
        for e in engs:
            e.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e, b in zip(engs, bufs):
            e.step_many(b.data_ptr(), n * esz, nbuf, steps, use_graph=True)
        for e in engs:
            e.sync()
        dt = time.perf_counter() - t0
        if r:
            best.append(dt / steps * 1e6)
    for e in engs:
        e.close()
    best.sort()
    return best[len(best) // 2], best[0]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--nbuf", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    for k in (1, 2, 4, 8, 1, 2):
        med, mn = run(a.kind, a.lanes, k, a.steps, a.nbuf, a.reps)
        print(f"kind {a.kind} lanes {a.lanes} partitions {k}: median {med:.3f} us/step  min {mn:.3f}", flush=True)
