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
            kwargs = {}
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

    def setup_stats_allreduce(self, prefer_native: bool = True) -> str:
        """Choose how the 4 statistics doubles are summed over ranks.  Preferred: the C ABI's own RCCL communicator
        (rank 0's ncclUniqueId travels through the torch.distributed store).  If the library's RCCL path cannot be
        set up, torch.distributed's all-reduce (the same RCCL underneath) is used and the reason is kept."""
        if not self.coll.active:
            return self.allreduce_path
        self.allreduce_path = f"torch.distributed({self.coll.backend})"
        if prefer_native and hasattr(self.engine, "comm_init"):
            # 1. every rank checks that it can reach the library's RCCL entry points at all (a rank that cannot must not
            #    leave the others waiting inside ncclCommInitRank); rank 0's id is the one that is used
            uid, err = None, None
            try:
                uid = self.engine.comm_unique_id()
            except Exception as exc:  # librccl missing on this rank
                err = repr(exc)
            if self.coll.sum([0.0 if err else 1.0])[0] != float(self.info.world):
                self.allreduce_note = f"native RCCL path unavailable on at least one rank: {err}"
                return self.allreduce_path
            # 2. rank 0's 128 bytes travel through the torch.distributed store; every rank joins
            uid = self.coll.broadcast_from_root(uid)
            ok = 0.0
            try:
                self.engine.comm_init(self.info.world, self.info.rank, uid)
                ok = 1.0
            except Exception as exc:
                err = repr(exc)
            # 3. all ranks or none: a communicator only part of the ranks hold must never see a collective
            if self.coll.sum([ok])[0] == float(self.info.world):
                self.allreduce_path = "gymrs_allreduce_stats (RCCL via the C ABI)"
            else:
                self.allreduce_note = f"native RCCL path unavailable: {err}"
        return self.allreduce_path

    @property
    def native(self) -> bool:
        return self.allreduce_path.startswith("gymrs_allreduce_stats")

    def allreduce_stats(self) -> np.ndarray:
        """{sum_return, sum_length, n_episodes, n_steps} of the WHOLE batch, identical on every rank."""
        if self.native:
            return np.asarray(self.engine.allreduce_stats(), dtype=np.float64)
        return self.coll.sum(self.engine.stats())

    def check_total_steps(self, total_stats: np.ndarray, steps_per_lane: int) -> None:
        want = float(self.total_lanes) * float(steps_per_lane)
        if float(total_stats[3]) != want:
            raise AssertionError(f"all-reduced n_steps {total_stats[3]} != lanes {self.total_lanes} x steps {steps_per_lane}")

    def job_rate(self, steps_per_lane: int, wall_seconds_max_over_ranks: float) -> float:
        """Whole-job env-steps/s: the units ALL ranks processed / the slowest rank's time."""
        return float(self.total_lanes) * float(steps_per_lane) / wall_seconds_max_over_ranks
