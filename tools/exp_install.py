"""After another engine has run for a while, an engine is slow for ~4000 steps (6.9 -> 6.45 us).  Is that its lines finding
their way into the Infinity Cache?  Try to install them at once: k steps with plain (allocating) accesses first."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")
n, nbuf, steps = 1 << 20, 32, 500
ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
engs = []
for i in range(2):
    e = gymrs.BatchedEngine(0, n, flags=3)
    e.reset(seed=1)
    for j in range(nbuf):
        e.fill_actions(ring[j].data_ptr(), seed=2, t=j)
    engs.append(e)
a, b = engs


def series(e, label, reps=10):
    ts = []
    for _ in range(reps):
        e.sync(); t0 = time.perf_counter()
        e.step_many(ring.data_ptr(), n, nbuf, steps)
        e.sync(); ts.append((time.perf_counter() - t0) / steps * 1e6)
    print(f"{label:44s}" + " ".join(f"{t:.2f}" for t in ts), flush=True)


series(a, "A warm-up", 16)
series(b, "B after A (library's own take-over step)")
series(a, "A after B (library's own take-over step)")
series(b, "B after A (library's own take-over step)")
series(b, "B after A (nothing done)")
series(a, "A after B (nothing done)")
for k in (1, 8, 64):
    b.set_tuning(4, 2)
    b.step_many(ring.data_ptr(), n, nbuf, k)
    b.set_tuning(4, 0)
    series(b, f"B after A, {k} plain step(s) first")
    series(a, "A after B (nothing done)")
big = torch.empty(1 << 27, dtype=torch.float32, device="cuda:0")  # 512 MiB
big.zero_(); big.add_(1.0); torch.cuda.synchronize()
series(b, "B after A and a 512 MiB torch sweep")
