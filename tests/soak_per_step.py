"""One-off long-run parity check of the PER-STEP path (a script next to the suite, not collected by pytest): 40 000 CartPole
steps -- 10 001 eager, 20 000 as HIP-graph replays, 9 999 eager -- of 4099 lanes with the reset log on (5 000 in-kernel
folds, 9.6 million episodes), global env ids beyond 2^33, a seed beyond 2^63: statistics and state bits must equal the
CPU f32 twin stepped one by one.
    gpurun -- 'python tests/soak_per_step.py'
"""
import importlib, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
gymrs = importlib.import_module("gym-rs_amd")
from oracle.bindings import Twin, TwinEngine
tw_lib = Twin()
n, steps, nbuf = 4099, 40000, 7
flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
eng = gymrs.BatchedEngine(0, n, flags=flags, global_env_offset=(1 << 33) + 5)
tw = TwinEngine(tw_lib, 0, n, gymrs.engine.default_params(0), flags=flags, gid0=(1 << 33) + 5)
seed = (1 << 63) + 12345
eng.reset(seed=seed); tw.reset(seed)
bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
acts = [tw.fill_actions(8, b) for b in range(nbuf)]
for b in range(nbuf): eng.fill_actions(bufs[b].data_ptr(), seed=8, t=b)
t0 = time.time()
done = 0
for chunk, graph in ((10001, False), (20000, True), (9999, False)):
    eng.step_many(bufs.data_ptr(), n, nbuf, chunk, use_graph=graph)
    for t in range(chunk): tw.step(acts[t % nbuf])
    done += chunk
    assert np.array_equal(eng.stats(), tw.stats()), (done, eng.stats(), tw.stats())
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32)), done
print("per-step soak ok:", done, "steps", eng.stats(), f"{time.time()-t0:.1f}s")

# The same with all three flags and a 45-step time limit that random-policy episodes reach now and then: launches with and
# without the limit alternate (GYMRS_TIME_LIMIT elision, docs/history/DESIGN_rounds_1-4.md 3.1 item 6b), refreshes of the bound with back-off,
# graph replays (which keep the limit) in the middle.
import json
p = gymrs.engine.default_params(0)
p.max_episode_steps = 45
flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | gymrs.TIME_LIMIT
eng = gymrs.BatchedEngine(0, n, flags=flags, params=p, global_env_offset=77)
tw = TwinEngine(tw_lib, 0, n, p, flags=flags, gid0=77)
eng.reset(seed=3); tw.reset(3)
t0 = time.time()
done = 0
for chunk, graph in ((9001, False), (6000, True), (9999, False)):
    eng.step_many(bufs.data_ptr(), n, nbuf, chunk, use_graph=graph)
    for t in range(chunk): tw.step(acts[t % nbuf])
    done += chunk
    assert np.array_equal(eng.stats(), tw.stats()), (done, eng.stats(), tw.stats())
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32)), done
    assert np.array_equal(eng.get_step_result()[2], tw.get_result()[2]), done
extra = json.loads(eng.env_json(0))["gymrs"]
assert 0 < extra["time_limit_elided_launches"] < done
print("time-limit soak ok:", done, "steps", eng.stats(), {k: v for k, v in extra.items() if k.startswith("time_limit")}, f"{time.time()-t0:.1f}s")
