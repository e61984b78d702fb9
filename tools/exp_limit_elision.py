"""Diagnostics for the GYMRS_TIME_LIMIT elision: us per step and how many launches ran without the limit."""
import importlib
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")

n, nbuf = 1 << 20, 32
for kind in (0, 1):
    for flags in (3, 7):
        eng = gymrs.BatchedEngine(kind, n, flags=flags)
        eng.reset(seed=1)
        bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
        for j in range(nbuf):
            eng.fill_actions(bufs[j].data_ptr(), seed=2, t=j)
        for steps in (3000, 3000, 3000):
            eng.sync()
            t0 = time.perf_counter()
            eng.step_many(bufs.data_ptr(), n, nbuf, steps)
            eng.sync()
            dt = (time.perf_counter() - t0) / steps * 1e6
            extra = json.loads(eng.env_json(0))["gymrs"]
            print(f"kind {kind} flags {flags}: {dt:.3f} us/step  elided {extra.get('time_limit_elided_launches')} refreshes {extra.get('time_limit_refreshes')} waits {extra.get('time_limit_waits')} wait_us {extra.get('time_limit_wait_us')} tick {extra['tick']}", flush=True)
        eng.close()
