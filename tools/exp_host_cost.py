"""Host cost of one eager step launch: how long gymrs_step_many takes to RETURN (launches queued, GPU still busy)."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")
for kind, n in ((0, 1 << 20), (1, 1 << 20), (0, 1 << 22)):
    e = gymrs.BatchedEngine(kind, n, flags=3)
    e.reset(seed=1)
    ring = torch.empty((8, n), dtype=torch.uint8, device="cuda:0")
    for j in range(8):
        e.fill_actions(ring[j].data_ptr(), seed=2, t=j)
    e.step_many(ring.data_ptr(), n, 8, 500)
    e.sync()
    for steps in (200, 1000, 1000):
        t0 = time.perf_counter()
        e.step_many(ring.data_ptr(), n, 8, steps)
        t1 = time.perf_counter()
        e.sync()
        t2 = time.perf_counter()
        print(f"kind {kind} n 2^{n.bit_length()-1} steps {steps}: call returned after {(t1-t0)/steps*1e6:.2f} us/step, finished after {(t2-t0)/steps*1e6:.2f} us/step", flush=True)
    e.close()
