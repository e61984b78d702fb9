"""The N>1 path on CPU (world_size 2, gloo): bench.py's own per-rank function -- sharding by global env offset,
repetitions, max-over-ranks, the statistics all-reduce after the clock, the line rank 0 prints -- with the f32 twin
standing in for the GPU shard (it supplies ONLY the per-shard stepping and statistics).  Plus the launcher pieces
(`python bench.py --gpus N` becoming N ranks) that do not need a device."""
import importlib
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

# A CPU stand-in for bench.HipBackend: same duck type, engines are f32 twins (tests may use oracle/).
TWIN_BACKEND = r'''
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle.bindings import Twin, TwinEngine


class _Ring:
    def __init__(self, bufs):
        self.bufs = bufs


class TwinShard:
    """The slice of BatchedEngine's surface that bench.run_rank touches."""

    def __init__(self, twin, gymrs, kind, n, offset, flags):
        self.params = gymrs.engine.default_params(kind)
        self._e = TwinEngine(twin, kind, n, self.params, flags=flags, gid0=offset)
        self.stream = 0
        self.obs_dim = {0: 4, 1: 2, 2: 3}[kind]
        self.global_env_offset = offset
        self.n_envs = n

    def reset(self, seed=None, options=None):
        self._e.reset(seed)

    def set_tuning(self, *a):
        pass

    def fill_actions(self, seed, t):
        return self._e.fill_actions(seed, t)

    def step_many(self, ring, stride, nbuf, k, use_graph=False):
        for t in range(k):
            self._e.step(ring.bufs[t % nbuf])

    def sync(self):
        pass

    def stats(self):
        return self._e.stats()

    def stats_clear(self):
        self._e.stats_clear()

    def get_state(self):
        return self._e.get_state()

    def close(self):
        pass


class TwinBackend:
    name = "twin"
    collective_backend = "gloo"
    device = None
    dev_index = None

    def __init__(self, gymrs):
        self.gymrs = gymrs
        self.twin = Twin()
        self.engines = []

    def make_engine(self, kind, n, offset, flags, vec):
        e = TwinShard(self.twin, self.gymrs, kind, n, offset, flags)
        self.engines.append(e)
        return e

    def make_action_ring(self, eng, n, nbuf, is_float):
        ring = _Ring([eng.fill_actions(1, b) for b in range(nbuf)])
        return ring, 0, ring

    def stream_of(self, eng):
        return None

    def mark(self, stream):
        return time.perf_counter()

    def elapsed_ms(self, a, b):
        return (b - a) * 1e3

    def sync(self):
        pass

    def cpu_baseline(self, kind, seconds):
        return {"value": 1.0, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "stub for the CPU test"}
'''

WORKER = TWIN_BACKEND + r'''
import importlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
gymrs = importlib.import_module("gym-rs_amd")
args = bench.parse_args(["--gpus", "2", "--steps", "7", "--warmup", "5", "--n-envs", "3001", "--action-buffers", "4",
                         "--cpu-seconds", "0", "--repetitions", "3", "--no-probe"])
bench.MIN_REPETITION_SECONDS = 0.0  # one pass of K steps per repetition keeps the twin run short
info = gymrs.sharded.rank_info()
assert info.world == 2 and info.launched
backend = TwinBackend(gymrs)
out = bench.run_rank(args, info, backend)
shard = backend.engines[0]
state = shard.get_state()
np.save(os.path.join(sys.argv[2], f"state{info.rank}.npy"), state)
if info.is_root:
    print("RESULT " + json.dumps(out))
else:
    assert out is None
'''


def test_bench_rank_logic_two_ranks_gloo(tmp_path, twin):
    gymrs = importlib.import_module("gym-rs_amd")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = gymrs.sharded.spawn_command(str(script), [str(ROOT), str(tmp_path)], 2)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(line) == 1, res.stdout[-2000:]
    out = json.loads(line[0][len("RESULT "):])
    n, steps, warmup, reps = 3001, 7, 5, 3
    passes = out["timing"]["passes_per_repetition"]
    assert out["n_gpus"] == 2 and out["config"]["total_lanes"] == 2 * n and out["config"]["lanes_per_gpu"] == n
    assert out["timing"]["repetitions"] == reps and len(out["timing"]["wall_ms_per_repetition"]) == reps
    assert [r["global_env_offset"] for r in out["ranks"]] == [0, n]
    assert out["config"]["stats_allreduce"] == "torch.distributed(gloo)"  # the twin has no RCCL communicator
    assert out["value"] == pytest.approx(2 * n * steps * passes / (out["ms_per_step"] * 1e-3 * steps * passes), rel=1e-9)
    # the same batch, UNSHARDED, same schedule: warm-up, one calibration pass, stats_clear, then the timed steps
    from oracle.bindings import TwinEngine

    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    full = TwinEngine(twin, 0, 2 * n, gymrs.engine.default_params(0), flags=flags, gid0=0)
    ring = [full.fill_actions(1, b) for b in range(4)]
    full.reset(0)

    def run(k):
        for t in range(k):
            full.step(ring[t % 4])

    run(warmup)
    for _ in range(out["timing"]["calibration_passes"]):  # bench's untimed calibration passes
        run(steps)
    full.stats_clear()
    for _ in range(reps * passes):
        run(steps)
    want = full.stats()
    got = out["episodes"]
    assert (got["sum_return"], got["sum_length"], got["n_episodes"]) == (want[0], want[1], want[2])
    assert want[3] == 2 * n * steps * passes * reps
    # shard invariance of the state itself: rank r's lanes are lanes [r*n, (r+1)*n) of the unsharded batch, bit for bit
    cat = np.concatenate([np.load(tmp_path / f"state{r}.npy") for r in range(2)], axis=1)
    assert np.array_equal(cat.view(np.uint32), full.get_state().view(np.uint32))


def test_launcher_pieces():
    gymrs = importlib.import_module("gym-rs_amd")
    sh = gymrs.sharded
    assert sh.needs_spawn(8, env={}) and not sh.needs_spawn(1, env={}) and sh.needs_spawn(1, env={}, force=True)
    assert not sh.needs_spawn(8, env={"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"})
    info = sh.rank_info({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"})
    assert (info.rank, info.local_rank, info.world, info.launched, info.is_root) == (3, 3, 8, True, False)
    assert sh.rank_info({}) == sh.RankInfo(0, 0, 1, False)
    assert sh.shard_offset(3, 1 << 20) == 3 << 20  # BASELINE configs[4]: rank r starts at r * 2^20
    cmd = sh.spawn_command("bench.py", ["--gpus", "8", "--steps", "20"], 8, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "8", "--steps", "20"]
    import bench

    assert bench.choose_passes(136e-6) == 37 and bench.choose_passes(0.02) == 1 and bench.choose_passes(0.0) == 1


def test_plain_multi_gpu_form_spawns_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` (the driver's plain form) must start two ranks itself; in this container they stop
    with the library's "no GPU" message instead of the round-1 behaviour (exit 2 before doing anything)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_bench_contract.py")
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    text = res.stdout + res.stderr
    assert res.returncode != 0
    assert "no GPU visible; the stepper has no CPU fallback" in text
    assert "torch.distributed" in text or "ChildFailedError" in text or "elastic" in text  # it went through the launcher


def test_rank_cpu_pinning_blocks():
    """sharded.pin_rank_to_cpus: a different block of 4 of the allowed CPUs per local rank, the previous mask handed back for
    restore_cpus; switched off by GYMRS_NO_CPU_PIN=1.  Run in a child process: the affinity of the test runner stays."""
    code = r'''
import importlib, json, os, sys
sys.path.insert(0, sys.argv[1])
sh = importlib.import_module("gym-rs_amd").sharded
allowed = sorted(os.sched_getaffinity(0))
out = {"allowed": allowed, "ranks": []}
for rank in (0, 1, 5):
    mine, before = sh.pin_rank_to_cpus(rank)
    out["ranks"].append({"mine": mine, "before": before, "now": sorted(os.sched_getaffinity(0))})
    sh.restore_cpus(before)
    assert sorted(os.sched_getaffinity(0)) == allowed
mine, before = sh.pin_rank_to_cpus(0, env={"GYMRS_NO_CPU_PIN": "1"})
out["off"] = [mine, sorted(os.sched_getaffinity(0)) == allowed]
print(json.dumps(out))
'''
    res = subprocess.run([sys.executable, "-c", code, str(ROOT)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    allowed = out["allowed"]
    if len(allowed) <= 4:
        assert all(r["mine"] is None for r in out["ranks"])  # nothing to choose from: left alone
        return
    blocks = len(allowed) // 4
    for rank, r in zip((0, 1, 5), out["ranks"]):
        start = (rank % blocks) * 4
        assert r["mine"] == allowed[start:start + 4] == r["now"] and r["before"] == allowed
    if blocks >= 2:
        assert out["ranks"][0]["mine"] != out["ranks"][1]["mine"]
    assert out["off"] == [None, True]
