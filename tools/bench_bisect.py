#!/usr/bin/env python
"""Developer tool: which part of bench.py's set-up changes the per-launch time?  Variants chosen by argv letters."""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
variant = sys.argv[1] if len(sys.argv) > 1 else ""
gymrs = importlib.import_module("gym-rs_amd")
n, nbuf, steps = 1 << 20, 32, 3000
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
eng = gymrs.BatchedEngine(0, n, global_env_offset=0, device=0, flags=flags)
stream = torch.cuda.ExternalStream(eng.stream, device=dev)
shape = (nbuf * n,) if "f" in variant else (nbuf, n)
ring = torch.empty(shape, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for b in range(nbuf):
    eng.fill_actions(ring.data_ptr() + b * n, seed=1, t=b)
eng.reset(seed=0)
def run(k):
    eng.step_many(ring.data_ptr(), n, nbuf, k)
run(500)
eng.sync()
if "c" in variant:
    run(steps); torch.cuda.synchronize()
if "s" in variant:
    eng.stats_clear()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    run(steps)
    e1.record(stream)
    if "g" in variant:
        eng.sync()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / steps)
print(f"variant '{variant}': " + " ".join(f"{t:.3f}" for t in ts), flush=True)
