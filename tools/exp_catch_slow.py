"""Catch a process in the slow mode (> 6.65 us per 2^20-lane CartPole step after warm-up) and see what one step with
plain (allocating) accesses does to it."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")
n, nbuf, steps = 1 << 20, 32, 500
ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
e = gymrs.BatchedEngine(0, n, flags=3)
e.reset(seed=1)
for j in range(nbuf):
    e.fill_actions(ring[j].data_ptr(), seed=2, t=j)


def series(reps):
    ts = []
    for _ in range(reps):
        e.sync(); t0 = time.perf_counter()
        e.step_many(ring.data_ptr(), n, nbuf, steps)
        e.sync(); ts.append((time.perf_counter() - t0) / steps * 1e6)
    return ts


ts = series(14)
tail = sorted(ts[6:])[4]
print("warm-up + 8 reps: " + " ".join(f"{t:.2f}" for t in ts), "SLOW" if tail > 6.65 else "fast", flush=True)
if tail > 6.65:
    e.set_tuning(4, 2); e.step_many(ring.data_ptr(), n, nbuf, 1); e.set_tuning(4, 0)
    print("  after ONE plain step: " + " ".join(f"{t:.2f}" for t in series(12)), flush=True)
    print("  ptrs", [hex(p) for p in e.state_ptrs()], hex(ring.data_ptr()), flush=True)
