"""One-off long-run parity check (too slow for the test-suite: the CPU twin needs ~40 s): 2*10^5 steps of every env
through the fused rollout kernel against the f32 twin stepped one by one -- global env ids beyond 2^32, a seed
beyond 2^63, time limit + auto-reset + statistics on.  State bits and integer statistics must be identical.
    gpurun -- 'python tests/soak_parity.py'
"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402

g = importlib.import_module("gym-rs_amd")
from oracle.bindings import Twin, TwinEngine  # noqa: E402

tw_lib = Twin()
for kind, n, steps in ((0, 4099, 200_000), (1, 2051, 200_000), (2, 1031, 100_000)):
    flags = g.AUTO_RESET | g.TRACK_STATS | g.TIME_LIMIT
    p = g.engine.default_params(kind)
    eng = g.BatchedEngine(kind, n, flags=flags, params=p, global_env_offset=4 * 10**9)
    tw = TwinEngine(tw_lib, kind, n, p, flags=flags, gid0=4 * 10**9)
    eng.reset(seed=2**63 + 17)
    tw.reset(2**63 + 17)
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        k = min(20_000, steps - done)
        eng.rollout(k, action_seed=77, action_t0=done)
        done += k
    eng.sync()
    t1 = time.perf_counter()
    for t in range(steps):
        tw.step(tw.fill_actions(77, t))
    t2 = time.perf_counter()
    same = np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    gs, ts = eng.stats(), tw.stats()
    print(f"kind {kind}: {n} lanes x {steps} steps, state bit-identical: {same}, stats {gs} vs {ts}, gpu {t1 - t0:.2f} s, twin {t2 - t1:.1f} s")
    assert same and np.array_equal(gs[1:], ts[1:]) and (gs[0] == ts[0] if kind != 2 else abs(gs[0] - ts[0]) <= 1e-6 * abs(ts[0]))
    eng.close()
print("SOAK OK")
