"""The N>1 path on CPU (world_size 2 and 8, gloo): bench.py's own per-rank function -- sharding by global env offset,
repetitions, max-over-ranks, the statistics all-reduce after the clock, the line rank 0 prints -- with the f32 twin
standing in for the GPU shard (it supplies ONLY the per-shard stepping and statistics).  Plus the launcher pieces
(`python bench.py --gpus N` becoming N ranks) that do not need a device."""
import importlib
import json
import os
import spawn_server
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
import importlib, json, os, sys, time
sys.path.insert(0, sys.argv[1])
import bench
gymrs = importlib.import_module("gym-rs_amd")
world, comm_mode = int(sys.argv[3]), sys.argv[4]
args = bench.parse_args(["--gpus", str(world), "--steps", "7", "--warmup", "5", "--n-envs", "3001", "--action-buffers", "4",
                         "--cpu-seconds", "0", "--repetitions", "3", "--no-probe"])
bench.MIN_REPETITION_SECONDS = 0.0  # one pass of K steps per repetition keeps the twin run short
bench.SETTLE_SECONDS = 0.0
info = gymrs.sharded.rank_info()
assert info.world == world and info.launched
backend = TwinBackend(gymrs)

if comm_mode != "none":
    # A stand-in for the C ABI's RCCL entry points (gymrs_comm_unique_id / _comm_init / _allreduce_stats): "ok" works (the sum
    # itself travels over gloo), the other modes misbehave on ONE rank the way a broken first contact would.
    import numpy as np

    class CommShard(TwinShard):
        is_probe = False

        def comm_unique_id(self):
            return b"u" * 128

        def comm_init(self, n_ranks, rank, uid):
            assert uid == b"u" * 128 and n_ranks == world
            if comm_mode == "init_hangs" and rank == 1 and self.is_probe:
                time.sleep(3600)
            if comm_mode == "init_fails" and rank == world - 1 and self.is_probe:
                raise RuntimeError("ncclCommInitRank: unhandled system error (test)")
            self.joined = True

        seq = 0

        def allreduce_stats(self):
            # the stand-in's own "wire" (files in the test's directory), independent of the torch.distributed group the way
            # the library's RCCL communicator is: a rank that never shows up leaves its peers waiting HERE, not in gloo
            assert self.joined
            if comm_mode == "trial_hangs" and info.rank == 0 and self.is_probe:
                time.sleep(3600)
            CommShard.seq += 1
            tag = ("p" if self.is_probe else "e") + str(CommShard.seq)
            mine = os.path.join(sys.argv[2], f"ar_{tag}_{info.rank}.npy")
            np.save(mine + ".tmp.npy", np.asarray(self.stats(), np.float64))
            os.replace(mine + ".tmp.npy", mine)
            total = np.zeros(4)
            for r in range(world):
                path = os.path.join(sys.argv[2], f"ar_{tag}_{r}.npy")
                while not os.path.exists(path):
                    time.sleep(0.01)
                total += np.load(path)
            return total

    def make_engine(kind, n, offset, flags, vec):
        e = CommShard(backend.twin, gymrs, kind, n, offset, flags)
        backend.engines.append(e)
        return e

    def probe_engine(kind, offset, flags):
        e = CommShard(backend.twin, gymrs, kind, 16, offset, flags)
        e.is_probe = True
        e.reset(0)
        return e

    backend.make_engine = make_engine
    backend.probe_engine = probe_engine

out = bench.run_rank(args, info, backend)
shard = backend.engines[0]
state = shard.get_state()
import numpy as np
np.save(os.path.join(sys.argv[2], f"state{info.rank}.npy"), state)
if info.is_root:
    print("RESULT " + json.dumps(out), flush=True)
else:
    assert out is None
if getattr(args, "hard_exit", False):  # a helper thread still sits in the stand-in's sleep
    sys.stdout.flush()
    os._exit(0)
'''


def run_workers(tmp_path, world, comm_mode="none", timeout=900, extra_env=None):
    gymrs = importlib.import_module("gym-rs_amd")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = gymrs.sharded.spawn_command(str(script), [str(ROOT), str(tmp_path), str(world), comm_mode], world)
    env = dict(os.environ, OMP_NUM_THREADS="1", **(extra_env or {}))
    res = spawn_server.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(line) == 1, res.stdout[-2000:]
    return json.loads(line[0][len("RESULT "):])


def check_against_one_unsharded_twin(tmp_path, twin, out, world, n=3001, steps=7, warmup=5, reps=3):
    gymrs = importlib.import_module("gym-rs_amd")
    passes = out["timing"]["passes_per_repetition"]
    assert out["n_gpus"] == world and out["config"]["total_lanes"] == world * n and out["config"]["lanes_per_gpu"] == n
    assert out["timing"]["repetitions"] == reps and len(out["timing"]["wall_ms_per_repetition"]) == reps
    assert [r["rank"] for r in out["ranks"]] == list(range(world))
    assert [r["global_env_offset"] for r in out["ranks"]] == [k * n for k in range(world)]
    for r in out["ranks"]:  # every rank's own launch time, with its spread over the repetitions
        assert 0 < r["launch_us_min"] <= r["launch_us"] <= r["launch_us_max"]
    ev = out["timing"]["event_us_per_step"]
    assert ev["min"] <= ev["median"] <= ev["max"]
    assert out["value"] == pytest.approx(world * n * steps * passes / (out["ms_per_step"] * 1e-3 * steps * passes), rel=1e-9)
    # the same batch, UNSHARDED, same schedule: warm-up, the calibration passes, stats_clear, then the timed steps
    from oracle.bindings import TwinEngine

    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    full = TwinEngine(twin, 0, world * n, gymrs.engine.default_params(0), flags=flags, gid0=0)
    ring = [full.fill_actions(1, b) for b in range(4)]
    full.reset(0)

    def run(k):
        for t in range(k):
            full.step(ring[t % 4])

    run(warmup)
    assert sum(out["timing"]["calibration_calls"]) == steps * out["timing"]["calibration_passes"]
    for k in out["timing"]["calibration_calls"]:  # bench's untimed calibration calls (the ring index restarts with every call)
        run(k)
    full.stats_clear()
    # a repetition is ONE step_many call of passes * steps steps; per call shape one uncounted lead-in repetition, and the second call
    # shape (the chain, reported separately) starts with one untimed call of a repetition's length
    assert set(out["paths"]) == {"per_step_visible", "chain"} and out["config"]["call_shape"] == "per_step_visible"
    calls = 2 * (reps + 1) + 1
    for _ in range(calls):
        run(steps * passes)
    want = full.stats()
    got = out["episodes"]
    assert (got["sum_return"], got["sum_length"], got["n_episodes"]) == (want[0], want[1], want[2])
    assert want[3] == world * n * steps * passes * calls
    assert out["value"] == out["paths"]["per_step_visible"]["value"] and out["roofline"] == out["paths"]["per_step_visible"]["roofline"]
    assert out["ranks_agree"] is True and out["expected_job_rate_from_rank_launch_times"] > 0
    for r in out["ranks"]:
        assert set(r["paths"]) == {"per_step_visible", "chain"}
    # shard invariance of the state itself: rank r's lanes are lanes [r*n, (r+1)*n) of the unsharded batch, bit for bit
    cat = np.concatenate([np.load(tmp_path / f"state{r}.npy") for r in range(world)], axis=1)
    assert np.array_equal(cat.view(np.uint32), full.get_state().view(np.uint32))


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rank_logic_gloo(tmp_path, twin, world):
    '''bench.run_rank itself with world_size 2 and 8 (BASELINE configs[4]'s shape: 8 shards, global env ids rank * n + i).'''
    out = run_workers(tmp_path, world)
    assert out["config"]["stats_allreduce"] == "torch.distributed(gloo)"  # the twin has no RCCL communicator
    assert out["config"]["control_plane"].startswith("torch.distributed(gloo)")
    assert "cpu_baseline" not in out  # --cpu-seconds 0
    check_against_one_unsharded_twin(tmp_path, twin, out, world)


@pytest.mark.parametrize("comm_mode", ["ok", "init_hangs", "init_fails", "trial_hangs"])
def test_first_contact_with_the_native_collective_is_watched(tmp_path, twin, comm_mode):
    '''a rank stuck inside ncclCommInitRank (or a first all-reduce whose peers never arrive) used to hang the
    job.  The native path is first tried on a throw-away probe engine under a watchdog; whatever goes wrong on ANY rank, ALL
    ranks fall back to torch.distributed, the line says why, and the statistics are still those of the unsharded batch.'''
    out = run_workers(tmp_path, 2, comm_mode, extra_env={"GYMRS_COMM_TIMEOUT": "3"})
    cfg = out["config"]
    if comm_mode == "ok":
        assert cfg["stats_allreduce"].startswith("gymrs_allreduce_stats") and "stats_allreduce_note" not in cfg
        assert cfg["comm_watchdog"] == "not triggered"
    else:
        assert cfg["stats_allreduce"] == "torch.distributed(gloo)"
        note = cfg["stats_allreduce_note"]
        assert "torch.distributed on ALL ranks" in note and "probe engine" in note
        if comm_mode == "trial_hangs":  # rank 0 (the one that prints) is the rank that gave up waiting
            assert "did not complete within 3 s" in note and cfg["comm_watchdog"].startswith("a native RCCL call timed out")
        if comm_mode == "init_fails":
            assert "failed" in note or "another rank" in note
    check_against_one_unsharded_twin(tmp_path, twin, out, 2)


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

    assert bench.choose_passes(136e-6, 5e-3) == 37 and bench.choose_passes(0.02, 5e-3) == 1 and bench.choose_passes(0.0) == 1
    assert bench.choose_passes(128e-6) * 128e-6 >= 0.1  # a repetition lasts >= 100 ms (profiles/r04_launches_per_call.log)


def test_plain_multi_gpu_form_spawns_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` (the driver's plain form) must start two ranks itself; in this container they stop
    with the library's "no GPU" message instead of the round-1 behaviour (exit 2 before doing anything)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_bench_contract.py")
    res = spawn_server.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0"],
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
    res = spawn_server.run([sys.executable, "-c", code, str(ROOT)], capture_output=True, text=True, timeout=120)
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


def test_rank_pinning_near_the_gpu_from_a_fake_topology(tmp_path):
    """sharded.pin_rank_near_gpu (VERDICT r2 "next" #1d): the block of CPUs comes from the GPU's own NUMA node -- KFD topology node
    -> PCI function -> local_cpulist -- ranks sharing a node take consecutive blocks, ROCR/HIP_VISIBLE_DEVICES re-index the GPUs,
    and anything unreadable falls back to the index-based blocks.  A fake /sys stands in for an 8-GPU, 2-socket host whose CPU
    numbers are this process's own allowed CPUs (so that sched_setaffinity accepts them)."""
    code = r'''
import importlib, json, os, sys
sys.path.insert(0, sys.argv[1])
sh = importlib.import_module("gym-rs_amd").sharded
root = sys.argv[2]
allowed = sorted(os.sched_getaffinity(0))
half = len(allowed) // 2
sockets = [allowed[:half], allowed[half:]]
def cpulist(c): return ",".join(str(x) for x in c)
nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
for i in range(2):  # two CPU nodes first, as KFD lists them
    os.makedirs(os.path.join(nodes, str(i)))
    open(os.path.join(nodes, str(i), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
for g in range(8):
    bus = 0x10 + 0x10 * g
    os.makedirs(os.path.join(nodes, str(2 + g)))
    open(os.path.join(nodes, str(2 + g), "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
    dev = os.path.join(root, "bus", "pci", "devices", f"0000:{bus:02x}:00.0")
    os.makedirs(dev)
    open(os.path.join(dev, "numa_node"), "w").write(f"{g // 4}\n")
    open(os.path.join(dev, "local_cpulist"), "w").write(cpulist(sockets[g // 4]) + "\n")
out = {"allowed": allowed, "map": sh.gpu_numa_map(root, env={}), "ranks": []}
for rank in range(8):
    mine, before, node = sh.pin_rank_near_gpu(rank, n_local_ranks=8, sysfs=root, env={})
    out["ranks"].append({"mine": mine, "node": node, "now": sorted(os.sched_getaffinity(0))})
    sh.restore_cpus(before)
out["visible"] = sh.gpu_numa_map(root, env={"HIP_VISIBLE_DEVICES": "5,1"})
out["uuid"] = sh.gpu_numa_map(root, env={"ROCR_VISIBLE_DEVICES": "GPU-abcdef"})
out["missing"] = sh.gpu_numa_map(os.path.join(root, "nope"), env={})
mine, before, node = sh.pin_rank_near_gpu(3, n_local_ranks=8, sysfs=os.path.join(root, "nope"), env={})
out["fallback"] = {"mine": mine, "node": node}
sh.restore_cpus(before)
mine, before, node = sh.pin_rank_near_gpu(3, n_local_ranks=8, sysfs=root, env={"GYMRS_NO_CPU_PIN": "1"})
out["off"] = [mine, node, sorted(os.sched_getaffinity(0)) == allowed]
print(json.dumps(out))
'''
    res = spawn_server.run([sys.executable, "-c", code, str(ROOT), str(tmp_path / "sys")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    allowed = out["allowed"]
    if len(allowed) < 8:
        pytest.skip("fewer than 8 CPUs allowed: nothing to partition")
    half = len(allowed) // 2
    sockets = [allowed[:half], allowed[half:]]
    assert [m[0] for m in out["map"]] == [0, 0, 0, 0, 1, 1, 1, 1] and out["map"][5][1] == sockets[1]
    width = max(1, min(4, half // 4))
    for rank, r in enumerate(out["ranks"]):
        node, k = rank // 4, rank % 4
        assert r["node"] == node and r["mine"] == sockets[node][k * width:(k + 1) * width] == r["now"], (rank, r)
    assert len({tuple(r["mine"]) for r in out["ranks"]}) == 8  # no two ranks share a block
    assert [m[0] for m in out["visible"]] == [1, 0]  # HIP_VISIBLE_DEVICES=5,1 -> GPU 5 (node 1), GPU 1 (node 0)
    assert out["uuid"] is None and out["missing"] is None
    assert out["fallback"]["node"] is None and out["fallback"]["mine"] is not None  # index-based block, as before
    assert out["off"] == [None, None, True]


def test_allreduce_stats_raises_instead_of_ending_the_process(monkeypatch):
    """ShardedRun.allreduce_stats used to call os._exit(3) on ANY failure of the native call.  Now an ordinary failure of the C ABI call is
    re-raised as it is, and a collective that never completes raises TimeoutError and marks the run abandoned (bench.py then leaves through its own
    hard exit after printing; a library user sees an exception)."""
    import time

    gymrs = importlib.import_module("gym-rs_amd")
    sharded = gymrs.sharded

    class Engine:
        def __init__(self, mode):
            self.mode = mode

        def allreduce_stats(self):
            if self.mode == "fails":
                raise gymrs.GymrsError(3, "ncclAllReduce: unhandled system error")
            time.sleep(5.0)
            return [0.0] * 4

        def stats(self):
            return [1.0, 2.0, 3.0, 4.0]

    class Coll:
        active = False

        def sum(self, x):
            return x

    info = sharded.RankInfo(rank=0, local_rank=0, world=1, launched=False)
    run = sharded.ShardedRun(info, 16, Coll(), lambda off, n: Engine("fails"))
    run.allreduce_path = "gymrs_allreduce_stats (RCCL via the C ABI)"
    with pytest.raises(gymrs.GymrsError, match="ncclAllReduce"):
        run.allreduce_stats()
    assert run.abandoned is False
    monkeypatch.setenv("GYMRS_COMM_TIMEOUT", "0.2")
    run = sharded.ShardedRun(info, 16, Coll(), lambda off, n: Engine("hangs"))
    run.allreduce_path = "gymrs_allreduce_stats (RCCL via the C ABI)"
    t0 = time.perf_counter()
    with pytest.raises(TimeoutError, match="did not complete"):
        run.allreduce_stats()
    assert run.abandoned is True and time.perf_counter() - t0 < 3.0
    # without the native path the statistics go through the collective the run was given
    run = sharded.ShardedRun(info, 16, Coll(), lambda off, n: Engine("fails"))
    assert list(run.allreduce_stats()) == [1.0, 2.0, 3.0, 4.0]
