"""How long do the slow phases of a pinned process last?  500-step repetitions for ~20 s; prints the runs of repetitions above 6.7 us."""
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
e.step_many(ring.data_ptr(), n, nbuf, 3000); e.sync()
ts = []
t_start = time.perf_counter()
duration = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
while time.perf_counter() - t_start < duration:
    e.sync(); t0 = time.perf_counter()
    e.step_many(ring.data_ptr(), n, nbuf, steps)
    e.sync(); ts.append(((time.perf_counter() - t0) / steps * 1e6, t0 - t_start))
slow = [t > 6.7 for t, _ in ts]
print(f"{len(ts)} repetitions, {sum(slow)} slow; median {sorted(t for t, _ in ts)[len(ts)//2]:.3f} us/step")
i = 0
while i < len(ts):
    if slow[i]:
        j = i
        while j < len(ts) and slow[j]:
            j += 1
        print(f"  slow phase at {ts[i][1]:.3f} s for {ts[j-1][1] - ts[i][1] + 0.0033:.3f} s ({j - i} repetitions, mean {sum(t for t, _ in ts[i:j]) / (j - i):.2f} us)")
        i = j
    else:
        i += 1
