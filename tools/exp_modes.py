"""Does the per-step time of the same engine configuration depend on WHERE its arrays were allocated?  k engines created
one after the other in one process (the earlier ones stay alive, so each gets fresh memory), each timed for r
repetitions; then the same engines timed again in reverse order."""
import argparse
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--engines", type=int, default=6)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--shared-ring", type=int, default=1)
a = ap.parse_args()
n, nbuf = 1 << 20, 32
ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
engs = []
for i in range(a.engines):
    e = gymrs.BatchedEngine(a.kind, n, flags=3)
    e.reset(seed=1)
    r = ring if a.shared_ring else torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for j in range(nbuf):
        e.fill_actions(r[j].data_ptr(), seed=2, t=j)
    engs.append((e, r))


def measure(e, r):
    ts = []
    for _ in range(a.reps):
        e.sync()
        t0 = time.perf_counter()
        e.step_many(r.data_ptr(), n, nbuf, a.steps)
        e.sync()
        ts.append((time.perf_counter() - t0) / a.steps * 1e6)
    print("   reps in order: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


for order in (range(a.engines), reversed(range(a.engines))):
    for i in order:
        e, r = engs[i]
        ptrs = e.state_ptrs() if hasattr(e, "state_ptrs") else None
        med, lo, hi = measure(e, r)
        print(f"engine {i}: median {med:.3f} min {lo:.3f} max {hi:.3f} us/step" + (f"  state[0] at {ptrs[0]:#x}" if ptrs else ""), flush=True)
