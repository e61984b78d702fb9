"""The launches that DO check the time limit, under a policy whose episodes reach it (emulated: weak gravity and force, so a
random policy survives): us per step with all three flags vs without the limit flag."""
import importlib, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
gymrs = importlib.import_module("gym-rs_amd")
n, nbuf = 1 << 20, 32
ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
for flags in (3, 7, 3, 7):
    p = gymrs.engine.default_params(0)
    p.gravity, p.force_mag, p.max_episode_steps = 0.2, 0.5, 500
    e = gymrs.BatchedEngine(0, n, flags=flags, params=p)
    e.reset(seed=1)
    for j in range(nbuf):
        e.fill_actions(ring[j].data_ptr(), seed=2, t=j)
    e.step_many(ring.data_ptr(), n, nbuf, 1600)  # past the first mass truncation at 500, 1000, 1500
    ts = []
    for _ in range(6):
        e.sync(); t0 = time.perf_counter()
        e.step_many(ring.data_ptr(), n, nbuf, 700)
        e.sync(); ts.append((time.perf_counter() - t0) / 700 * 1e6)
    s = e.stats()
    extra = json.loads(e.env_json(0))["gymrs"]
    print(f"flags {flags}: " + " ".join(f"{t:.2f}" for t in ts) + f"  mean episode length {s[1] / max(s[2], 1):.0f}  elided {extra.get('time_limit_elided_launches')} of {extra['tick']}", flush=True)
    e.close()
