"""What makes the first few thousand steps of an engine slower?  Time series of 1000-step repetitions around events:
a reset, a flush of the Infinity Cache by a 1 GiB device copy, a long idle, another engine's run."""
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")
n, nbuf, steps = 1 << 20, 32, 1000
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
e = gymrs.BatchedEngine(0, n, flags=3)
e.set_tuning(4, nt)
e.reset(seed=1)
for j in range(nbuf):
    e.fill_actions(ring[j].data_ptr(), seed=2, t=j)
big_a = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")
big_b = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")


def series(label, reps=8):
    ts = []
    for _ in range(reps):
        e.sync()
        t0 = time.perf_counter()
        e.step_many(ring.data_ptr(), n, nbuf, steps)
        e.sync()
        ts.append((time.perf_counter() - t0) / steps * 1e6)
    print(f"{label:34s}" + " ".join(f"{t:.2f}" for t in ts), flush=True)


series("after creation + reset")
series("continuing")
e.reset(seed=2)
series("after reset()")
big_b.copy_(big_a)
torch.cuda.synchronize()
series("after a 1 GiB device copy")
time.sleep(0.5)
series("after 0.5 s idle")
e.stats()
series("after stats()")
c = e.clone()
series("after clone() (other arrays written)")
c.step_many(ring.data_ptr(), n, nbuf, 3000)
c.sync()
series("after 3000 steps of the clone")
