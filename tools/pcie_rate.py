"""PCIe-inclusive step rate: the boundary hands over HOST action buffers (gymrs_step_host: copy, then step).
DESIGN.md section 4 quotes this figure; it is never bench.py's `value`."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401  (shares the HIP runtime)

g = importlib.import_module("gym-rs_amd")
n = 1 << 20
eng = g.BatchedEngine(g.CARTPOLE, n, flags=g.AUTO_RESET | g.TRACK_STATS)
eng.reset(seed=0)
rng = np.random.default_rng(0)
bufs = [rng.integers(0, 2, n).astype(np.uint8) for _ in range(8)]
for b in bufs:
    eng.step_host(b)
eng.sync()
K = 500
t0 = time.perf_counter()
for k in range(K):
    eng.step_host(bufs[k % 8])
eng.sync()
dt = (time.perf_counter() - t0) / K
print(f"step_host (pageable host buffer, 1 MiB of u8 actions per step): {dt * 1e6:.1f} us per step = {n / dt:.3e} env-steps/s")
eng.close()
