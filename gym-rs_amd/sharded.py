"""Lane-sharded runs over the GPUs of one node (BASELINE.json configs[4]).

The path shards trivially: lanes are independent (no cross-lane term in cartpole.rs:398-483 /
mountain_car.rs:398-435), so rank r owns the contiguous global env ids ``[r * n, (r + 1) * n)`` and the
global id feeds the Philox counter -- N lanes on one GPU are bit-identical to the concatenation of k
shards.  There is NO data-path collective.  What ranks exchange:

  * one all-reduce(sum) of the 4 statistics doubles (32 B per rank; latency-bound, xGMI link bandwidth is
    irrelevant) -- through the C ABI's own RCCL communicator (``gymrs_comm_init`` /
    ``gymrs_allreduce_stats``) on the GPU, or through ``torch.distributed`` when that is not available;
  * barriers and a max-over-ranks of the measured times (``torch.distributed``; "nccl" = RCCL on ROCm).

One process per GPU.  ``spawn_ranks`` turns a plain ``python script.py --gpus N`` into that shape by
re-running the script under ``torch.distributed.run``; ``rank_info`` reads what the launcher exported.
Everything here is host logic and runs unchanged on CPU with the gloo backend (tests/test_sharded_cpu.py
drives it with the f32 twin standing in for the GPU shard).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np


@dataclass(frozen=True)
class RankInfo:
    rank: int
    local_rank: int
    world: int
    launched: bool  # True when a launcher (torch.distributed.run) exported RANK/WORLD_SIZE

    @property
    def is_root(self) -> bool:
        return self.rank == 0


def rank_info(env=None) -> RankInfo:
    env = os.environ if env is None else env
    launched = "RANK" in env and "WORLD_SIZE" in env
    return RankInfo(int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", env.get("RANK", "0"))),
                    int(env.get("WORLD_SIZE", "1")), launched)


def shard_offset(rank: int, lanes_per_rank: int) -> int:
    """Weak scaling: every rank holds ``lanes_per_rank`` lanes, rank r the global ids from r * lanes_per_rank."""
    if rank < 0 or lanes_per_rank <= 0:
        raise ValueError("rank must be >= 0 and lanes_per_rank > 0")
    return rank * lanes_per_rank


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def needs_spawn(n_gpus: int, env=None, force: bool = False) -> bool:
    """A plain ``python script.py --gpus N`` (no launcher in the environment) with N > 1 has to start its ranks."""
    info = rank_info(env)
    return (not info.launched) and (n_gpus > 1 or force)


def spawn_command(script: str, argv: Sequence[str], n_gpus: int, port: Optional[int] = None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def spawn_ranks(script: str, argv: Sequence[str], n_gpus: int, port: Optional[int] = None, env=None) -> int:
    """Run ``script argv`` as ``n_gpus`` ranks (one process per GPU) and return the launcher's exit code.
    stdout/stderr are inherited, so rank 0's JSON line reaches the caller's stdout unchanged."""
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(spawn_command(script, argv, n_gpus, port), env=env)


def pin_rank_to_cpus(local_rank: int, width: int = 4, env=None, n_local_ranks: int = 1):
    """Pin the calling process to ``width`` consecutive CPUs of those it may use, a different block per local rank, and
    return (the new CPU list, the previous one) -- or (None, previous) when pinning is switched off
    (``GYMRS_NO_CPU_PIN=1``) or not possible.  Call it BEFORE the HIP runtime starts: its helper threads inherit the mask.

    Why: a per-step launch costs the launching thread 2.8 us when it stays on its core and 3.1-5.4 us when the scheduler
    moves it around (tools/exp_host_cost.py); a 2^20-lane step lasts 4 (MountainCar) to 6.4 us (CartPole): little slack.
    Left to the scheduler on a 256-CPU host, one process in two ran 2-6 % slower (6.5-6.9 instead of 6.4 us per step;
    MountainCar's 4 us launches 4.2-5.2 instead of 3.95), whatever the socket; confined to a few cores every process
    measured the fast figure (profiles/r02_cpu_pinning.log).  Which NUMA node the block lies on made no difference."""
    env = os.environ if env is None else env
    try:
        previous = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None, None
    if n_local_ranks > 1:  # few CPUs for many ranks: narrower blocks rather than shared ones
        width = max(1, min(width, len(previous) // int(n_local_ranks)))
    if env.get("GYMRS_NO_CPU_PIN") == "1" or len(previous) <= width:
        return None, previous
    blocks = len(previous) // width
    start = (int(local_rank) % blocks) * width
    mine = previous[start:start + width]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None, previous
    return mine, previous


def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_map(sysfs: str = "/sys", env=None):
    """[(numa_node, [cpus local to it])] for the GPUs a HIP process would enumerate, in HIP's order, WITHOUT starting the HIP
    runtime (its helper threads inherit the affinity mask set before it starts): the KFD topology nodes with a GPU, in node
    order (the order ROCr and HIP enumerate them in), mapped to their PCI function's numa_node / local_cpulist, then filtered
    by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when those are plain index lists.  None when the topology cannot be read
    (no KFD, a container without /sys/class/kfd): the caller falls back to index-based blocks."""
    env = os.environ if env is None else env
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(d) for d in os.listdir(base) if d.isdigit()))
    except OSError:
        return None
    gpus = []
    for nd in nodes:
        props = {}
        try:
            with open(os.path.join(base, str(nd), "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            return None
        if int(props.get("simd_count", "0")) == 0:  # a CPU node
            continue
        try:
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (KeyError, ValueError):
            return None
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        dev = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        try:
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read().strip())
            with open(os.path.join(dev, "local_cpulist")) as f:
                cpus = _parse_cpulist(f.read())
        except (OSError, ValueError):
            return None
        gpus.append((node, cpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):  # applied in this order by the runtimes
        val = env.get(var)
        if val:
            try:
                gpus = [gpus[int(i)] for i in val.split(",") if i.strip() != ""]
            except (ValueError, IndexError):
                return None  # UUIDs or out-of-range entries: do not guess
    return gpus or None


def pin_rank_near_gpu(local_rank: int, width: int = 4, env=None, n_local_ranks: int = 1, sysfs: str = "/sys"):
    """pin_rank_to_cpus, but the block is taken from the CPUs LOCAL to the rank's GPU (its PCI function's NUMA node): with 8
    launch threads on a 2-socket host a thread on the far socket pays a cross-socket hop for every doorbell and every
    mapped-memory word.  Ranks whose GPUs share a node take consecutive blocks of that node's CPUs.  Returns
    (cpus, previous, numa_node); falls back to pin_rank_to_cpus (numa_node None) when the topology is not readable or the
    node's CPUs are not in this process's mask."""
    env = os.environ if env is None else env
    gpus = None if env.get("GYMRS_NO_NUMA_PIN") == "1" else gpu_numa_map(sysfs, env)
    try:
        previous = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None, None, None
    if env.get("GYMRS_NO_CPU_PIN") == "1":
        return None, previous, None
    if gpus and 0 <= local_rank < len(gpus):
        node, local = gpus[local_rank]
        usable = [c for c in local if c in set(previous)]
        sharing = [r for r in range(min(n_local_ranks, len(gpus))) if gpus[r][0] == node]  # local ranks whose GPU sits on this node
        if usable and local_rank in sharing:
            w = max(1, min(width, len(usable) // max(1, len(sharing))))
            start = sharing.index(local_rank) * w
            mine = usable[start:start + w]
            if len(mine) == w:
                try:
                    os.sched_setaffinity(0, mine)
                    return mine, previous, node
                except OSError:
                    pass
    mine, previous = pin_rank_to_cpus(local_rank, width=width, env=env, n_local_ranks=n_local_ranks)
    return mine, previous, None


def restore_cpus(previous) -> None:
    if previous:
        try:
            os.sched_setaffinity(0, previous)
        except OSError:
            pass


class Collective:
    """Barrier, max-over-ranks and the statistics sum of a sharded run.  ``backend`` is "nccl" (= RCCL) on GPUs,
    "gloo" on CPU; with one rank and no launcher nothing is initialised and every call is the identity."""

    def __init__(self, info: RankInfo, backend: str, device=None, force: bool = False):
        self.info = info
        self.backend = backend
        self.device = device
        self.active = info.world > 1 or (force and info.launched)
        self._dist = None
        if self.active:
            import torch.distributed as dist

            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend == "gloo" and os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
                # a one-node rendezvous on the loopback address: gloo would otherwise look the container's hostname up first
                # (it may not resolve, or only after a DNS timeout) before falling back to the loopback interface
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            import datetime

            # a rendezvous or collective whose peers never arrive ends the job with an error instead of hanging it
            kwargs = {"timeout": datetime.timedelta(seconds=int(os.environ.get("GYMRS_DIST_TIMEOUT", "300")))}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend, **kwargs)
            self._dist = dist

    def _tensor(self, values, dtype=None):
        import torch

        t = torch.tensor(list(values), dtype=dtype or torch.float64)
        return t.to(self.device) if (self.backend == "nccl" and self.device is not None) else t

    def barrier(self) -> None:
        if self.active:
            self._dist.barrier()

    def max(self, values: Sequence[float]):
        if not self.active:
            return [float(v) for v in values]
        t = self._tensor(values)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]

    def sum(self, values: Sequence[float]) -> np.ndarray:
        if not self.active:
            return np.asarray(values, dtype=np.float64)
        t = self._tensor(values)
        self._dist.all_reduce(t)
        return t.cpu().numpy().astype(np.float64)

    def gather_to_root(self, obj):
        """Small python objects (per-rank timings) to rank 0; None elsewhere."""
        if not self.active:
            return [obj]
        out = [None] * self.info.world if self.info.is_root else None
        self._dist.gather_object(obj, out, dst=0)
        return out

    def broadcast_from_root(self, obj):
        if not self.active:
            return obj
        box = [obj if self.info.is_root else None]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def close(self) -> None:
        if self.active and self._dist is not None:
            self._dist.destroy_process_group()
            self.active = False


def call_with_timeout(fn, seconds: float):
    """Run fn() in a daemon thread; (True, result) when it returned in time, (False, exception or None) otherwise.  A native
    call that never returns (a rank stuck inside ncclCommInitRank or a collective whose peers never arrive) cannot be
    cancelled -- the thread is left behind and the caller goes on without whatever it was setting up."""
    import threading

    box = {}

    def work():
        try:
            box["result"] = fn()
        except BaseException as exc:  # noqa: BLE001 -- reported to the caller
            box["error"] = exc

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return False, None
    if "error" in box:
        return False, box["error"]
    return True, box.get("result")


class ShardedRun:
    """One rank's shard of a lane-sharded batch plus the job-level bookkeeping.

    ``make_engine(global_env_offset, n_lanes)`` returns the rank's engine: anything with ``stats()`` and, for
    the native RCCL path, ``comm_unique_id() / comm_init(world, rank, id) / allreduce_stats()`` (BatchedEngine).
    """

    def __init__(self, info: RankInfo, lanes_per_rank: int, coll: Collective, make_engine: Callable):
        self.info = info
        self.coll = coll
        self.lanes_per_rank = int(lanes_per_rank)
        self.offset = shard_offset(info.rank, self.lanes_per_rank)
        self.total_lanes = self.lanes_per_rank * info.world
        self.engine = make_engine(self.offset, self.lanes_per_rank)
        self.allreduce_path = "none (one rank)"
        self.allreduce_note = None
        self.abandoned = False  # a native RCCL call timed out and may still be stuck in a helper thread
        self._leaked = []

    def setup_stats_allreduce(self, prefer_native: bool = True, make_probe_engine: Optional[Callable] = None,
                              timeout: Optional[float] = None) -> str:
        """Choose how the 4 statistics doubles are summed over ranks.  Preferred: the C ABI's own RCCL communicator
        (rank 0's ncclUniqueId travels through the torch.distributed store).  If the library's RCCL path cannot be
        set up, torch.distributed's all-reduce (the run's control-plane backend: gloo in bench.py) is used and the reason is kept.

        First contact is guarded (a rank stuck inside ncclCommInitRank used to hang the job): every native
        step runs under a watchdog of `timeout` seconds (GYMRS_COMM_TIMEOUT, default 90), and before the run's own engine
        joins a communicator a throw-away PROBE engine (make_probe_engine: a few lanes, its own stream) does comm_init + one
        all-reduce -- a collective that never completes blocks the stream it was enqueued on for good, which must not be the
        stream the benchmark runs on.  Any rank failing or timing out anywhere -> ALL ranks use torch.distributed (agreed
        through torch.distributed itself), `allreduce_note` says why, and `abandoned` tells the caller that a native call may
        still be stuck in a helper thread (leave with os._exit once the result is out)."""
        if not self.coll.active:
            return self.allreduce_path
        self.allreduce_path = f"torch.distributed({self.coll.backend})"
        if not (prefer_native and hasattr(self.engine, "comm_init")):
            return self.allreduce_path
        if timeout is None:
            timeout = float(os.environ.get("GYMRS_COMM_TIMEOUT", "90"))
        world = float(self.info.world)

        def all_ok(ok: bool) -> bool:
            return self.coll.sum([1.0 if ok else 0.0])[0] == world

        def join(engine, what: str):
            """comm_unique_id on rank 0 -> broadcast -> comm_init on every rank (watched).  Returns (ok on ALL ranks, local error)."""
            uid, err = None, None
            try:  # every rank checks that it can reach the library's RCCL entry points at all; rank 0's id is the one used
                uid = engine.comm_unique_id()
            except Exception as exc:  # librccl missing on this rank
                err = f"{what}: comm_unique_id: {exc!r}"
            if not all_ok(err is None):
                return False, err or f"{what}: RCCL entry points unavailable on another rank"
            uid = self.coll.broadcast_from_root(uid)
            done, res = call_with_timeout(lambda: engine.comm_init(self.info.world, self.info.rank, uid), timeout)
            if not done:
                err = f"{what}: comm_init " + (f"failed: {res!r}" if res is not None else f"did not return within {timeout:.0f} s")
                self.abandoned = self.abandoned or res is None
            # all ranks or none: a communicator only part of the ranks hold must never see a collective
            if not all_ok(done):
                return False, err or f"{what}: comm_init failed or timed out on another rank"
            return True, None

        def trial(engine, what: str):
            done, res = call_with_timeout(engine.allreduce_stats, timeout)
            err = None
            if not done:
                err = f"{what}: all-reduce " + (f"failed: {res!r}" if res is not None else f"did not complete within {timeout:.0f} s")
                self.abandoned = self.abandoned or res is None
            if not all_ok(done):
                return False, err or f"{what}: all-reduce failed or timed out on another rank"
            return True, None

        if make_probe_engine is not None:
            probe = make_probe_engine(self.offset)
            ok, err = join(probe, "probe engine")
            if ok:
                ok, err = trial(probe, "probe engine")
            if not ok:
                self.allreduce_note = f"native RCCL path unavailable ({err}); statistics summed by torch.distributed on ALL ranks"
                self._leaked.append(probe)  # never closed: its stream may hold a collective that cannot finish
                return self.allreduce_path
            probe.close()
        ok, err = join(self.engine, "engine")
        if ok:
            self.allreduce_path = "gymrs_allreduce_stats (RCCL via the C ABI)"
        else:
            self.allreduce_note = f"native RCCL path unavailable ({err}); statistics summed by torch.distributed on ALL ranks"
        return self.allreduce_path

    @property
    def native(self) -> bool:
        return self.allreduce_path.startswith("gymrs_allreduce_stats")

    def allreduce_stats(self) -> np.ndarray:
        """{sum_return, sum_length, n_episodes, n_steps} of the WHOLE batch, identical on every rank."""
        if self.native:
            # watched: a native collective that never completes ends this rank loudly (the launcher then ends the others)
            timeout = float(os.environ.get("GYMRS_COMM_TIMEOUT", "90"))
            done, res = call_with_timeout(self.engine.allreduce_stats, timeout)
            if not done and res is not None:
                raise res  # an ordinary failure of the C ABI call (a status, a message): the caller decides (this used to end the process)
            if not done:
                # a collective that never completes cannot be cancelled, and its helper thread holds the engine's stream: the rank is marked
                # abandoned (bench.py then leaves through os._exit after printing; a library user sees the exception) instead of exiting HERE
                self.abandoned = True
                raise TimeoutError(f"gymrs sharded: rank {self.info.rank}: gymrs_allreduce_stats did not complete within {timeout:.0f} s "
                                   f"(a peer never arrived?); the communicator of this engine is unusable from now on")
            return np.asarray(res, dtype=np.float64)
        return self.coll.sum(self.engine.stats())

    def check_total_steps(self, total_stats: np.ndarray, steps_per_lane: int) -> None:
        want = float(self.total_lanes) * float(steps_per_lane)
        if float(total_stats[3]) != want:
            raise AssertionError(f"all-reduced n_steps {total_stats[3]} != lanes {self.total_lanes} x steps {steps_per_lane}")

    def job_rate(self, steps_per_lane: int, wall_seconds_max_over_ranks: float) -> float:
        """Whole-job env-steps/s: the units ALL ranks processed / the slowest rank's time."""
        return float(self.total_lanes) * float(steps_per_lane) / wall_seconds_max_over_ranks
