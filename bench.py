#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched stepper (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # or under an external launcher
    python bench.py --in-process --gpus N ...                       # ONE process, the C ABI's native sharder (gymrs_sharded_*)

What is printed: rank 0 writes ONE line to stdout, a digest of at most 4 KB (`compact_line`: the contract's fields, `config`, the headline's `roofline`,
`cpu_baseline`, one small record per other call shape and BASELINE config -- names and numbers, no prose); the COMPLETE record (per-repetition times, per-rank
records, notes, probes) goes to `bench_full.json` next to this script (`--full-out`) and, pretty-printed, to stderr; the line names the file (`full`).  Round 4
printed the complete record as the line: 35 KB, of which the driver kept the last 8 KB and recorded `parsed: null`.

A "step" is ONE Env::step() of every lane of the workload = one launch of the step kernel over the rank's block (auto-reset and statistics on).  TWO call shapes
are timed by the default run:
  per_step_visible  the headline (`value`, `ms_per_step`, `roofline`): every step's arrays are RELEASED when its launch ends, so a reader between two steps -- the
                    policy of examples/cartpole.rs:18-26, which looks at every step's ActionReward -- sees them: K launches through HIP (what a gymrs_step loop
                    enqueues; SURVEY H1 "keep the per-step round trip").  `roofline.queue_launch_us`: the same shape once more through the engine's own queue
                    with HIP's header on every packet (GYMRS_AQL=2), beside the HIP figure, never instead of it;
  chain             reported separately: gymrs_step_many's chain through the engine's own dispatcher -- nothing is released until the chain ends, the state lives
                    in the L2s in between (a multi-step-in-cache variant: SURVEY H1 "report separately").
`roofline` (roofline_of): `frac` = bytes the kernel moves BY CONSTRUCTION / launch time / peak, `frac_moved` = device-wide counter bytes / time / 8 TB/s,
`frac_counted` = the contract's algorithmic bytes (38 / 22 / 37 per env-step) -- the same three keys with the same meaning on every leg.

Workload at N=1: BASELINE.json configs[1], CartPole-v1 at 2^20 parallel envs, f32.  At N>1 every rank holds 2^20 lanes (weak scaling; configs[4] at N=8 is 2^23
lanes) with global env ids rank*2^20+i and no data-path collective; the only collective of the path is one RCCL all-reduce of the 4 statistics doubles, taken
AFTER the timed region (its cost is reported separately as stats_readout_us).  `sharder` says which form ran: "single engine", "process-per-gpu", "in-process".

Timing (">= 5 repetitions, report the median"): after W warm-up steps and ~60 ms of untimed stepping (the settle phase: a fresh process steps faster
for its first ~20 ms than it does in the long run) the bench times R = 9 repetitions and reports their median (min / max in the full record).  One repetition =
ONE gymrs_step_many call of P x K steps, bracketed by barrier + torch.cuda.synchronize() on both sides (wall clock) and by HIP events on the engine's stream
(launch time); P is chosen once so that a repetition lasts >= 100 ms (MIN_REPETITION_SECONDS says why).  Per repetition the MAX over ranks is taken, then the
MEDIAN over repetitions; ms_per_step = that time / (P * K), value = all lanes * P * K / that time.  Nothing but the K-step passes sits inside a timed repetition.

Inputs are resident in HBM before the timed region: the state arrays, and a ring of pre-generated random-policy action buffers (the `rng.gen_range(0..=1)` of
examples/cartpole.rs:19, produced on the device by the Philox action stream).  See DESIGN.md section 4.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import importlib.util
import json
import math
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENVS = {
    # name: (kind, default lanes per GPU, algorithmic bytes per env-step read / written, workload label)
    "cartpole": (0, 1 << 20, 17, 21, "CartPole-v1 @ 2^20 envs per GPU, f32, auto-reset, random policy"),
    "mountain_car": (1, 1 << 20, 9, 13, "MountainCar-v0 @ 2^20 envs per GPU, f32, auto-reset, random policy"),
    "pendulum": (2, 1 << 22, 12, 25, "Pendulum-v1 (spec-derived) @ 2^22 envs per GPU, f32, auto-reset, 200-step time limit, random policy"),
}
# Bytes per env-step the kernels MOVE by construction when nothing is served by a cache: MountainCar's constant reward store is elided (22 - 4),
# Pendulum's theta_dot observation column IS the state column and its flags are rewritten only when they change (37 - 4 - ~0.7)
MOVED_BYTES = {"cartpole": 38.0, "mountain_car": 18.0, "pendulum": 32.3}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
L2_PEAK_GBPS = 34500.0  # aggregate of the eight 4 MiB L2s, same guide ("L2 (per XCD)": ~34.5 TB/s): what bounds a cache-resident chain
# The two call shapes of the per-step API (docstring above): name -> (GYMRS_AQL for the calls, what it is)
PATHS = {
    "per_step_visible": ("0", "K launches through HIP, each with an agent-scope acquire and an end-of-kernel RELEASE (L2 write-back): every step's "
                              "observations / rewards / flags are visible to whatever runs between two steps -- what a gymrs_step loop enqueues"),
    "chain": ("1", "one gymrs_step_many chain through the engine's own HSA queue: agent-scope acquire on every launch, release only at the END of "
                   "the chain -- nobody can read a step's arrays before the chain ends (reported separately from the headline)"),
}
HEADLINE_PATH = "per_step_visible"
# VALU issue roofline of the fused rollout kernel: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction
VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2
# A timed repetition = ONE gymrs_step_many call of P x K launches lasting at least this long.  Every repetition starts after an idle gap (barrier, synchronise,
# events: ~0.1 ms), and for its first ~13 ms the device then runs ~1.7 % slower than in the long run (profiles/r04_launches_per_call.log: 800-2500 launches per call
# 6.44-6.50 us each, 8000: 6.39, 20 000: 6.365): repetitions of 5 ms (rounds 2-4) measured mostly that recovery, and the driver's form (--steps 20) read 2 % lower
# than the default form (--steps 1000) on every box.  100 ms repetitions report the sustained rate whatever K the command line names.
MIN_REPETITION_SECONDS = 100e-3
REPETITIONS = 9
# Untimed stepping between the warm-up and the first timed repetition.  A process that has just started steps FASTER for its
# first ~20 ms than it does in the long run (profiles/r03_slow_mode.log: 6.31 us per step 15 ms in, 6.46 from ~40 ms on, on
# the same box; the memory side relaxes -- the same-footprint copy probe goes 5.1 -> 5.6 us while the shader clock and a
# VALU-only kernel do not move), so a bench that times 25 ms right after a 25-step warm-up reports a drifting transient
# (round 2's driver record: 6.43 -> 6.96 us over its 5 repetitions).  The sustained rate is the honest one.
SETTLE_SECONDS = 60e-3
# The other BASELINE.json configs, measured by the default N=1 run after the headline and attached as `configs`
# (VERDICT r2 "next" #3): name -> (env, lanes, action buffers)
EXTRA_CONFIGS = {
    "mountain_car_2p20": ("mountain_car", 1 << 20, 32),
    "pendulum_2p22": ("pendulum", 1 << 22, 32),         # BASELINE configs[3] as the stand-alone `--env pendulum` run has it
    "pendulum_2p22_8_action_buffers": ("pendulum", 1 << 22, 8),  # the same with a ring that fits the Infinity Cache (print both)
    "cartpole_2p24_dram_resident": ("cartpole", 1 << 24, 8),
    "cartpole_2p25_hbm_streaming": ("cartpole", 1 << 25, 8),  # 1.27 GB per step: what a step READS (544 MiB) no longer fits the Infinity Cache either
}


def kernel_source_sha16() -> str:
    """Identifies the kernels a PMC figure under profiles/ was collected with: traffic measured with other
    sources is stale and must not be printed beside a new kernel's time."""
    h = hashlib.sha256()
    for p in sorted((ROOT / "gym-rs_amd" / "csrc").glob("gymrs_*")):
        if p.suffix in (".h", ".hip"):
            h.update(p.name.encode())
            h.update(p.read_bytes())
    return h.hexdigest()[:16]


class submission:
    """The call shape of the gymrs_step_many calls inside the block (PATHS): the library looks GYMRS_AQL up per call."""

    def __init__(self, path):
        self.value = PATHS[path][0] if path in PATHS else None

    def __enter__(self):
        self.before = os.environ.get("GYMRS_AQL")
        if self.value is not None:
            os.environ["GYMRS_AQL"] = self.value

    def __exit__(self, *exc):
        if self.value is None:
            return
        if self.before is None:
            os.environ.pop("GYMRS_AQL", None)
        else:
            os.environ["GYMRS_AQL"] = self.before


def load_free_running_traffic(config_name: str, path: str, sha: str):
    """Fabric bytes per launch of FREE-RUNNING launches on this call shape: device-wide counter sampling around an undisturbed run
    (tools/devcount: rocprofiler-sdk's device counting service; rocprofv3 --pmc counts per dispatch and serialises the queues), kept
    in profiles/devcount_traffic.json together with the hash of the kernel sources it was taken with."""
    try:
        data = json.loads((ROOT / "profiles" / "devcount_traffic.json").read_text())
    except Exception:
        return None, "profiles/devcount_traffic.json missing"
    if data.get("kernel_source_sha16") != sha:
        return None, (f"profiles/devcount_traffic.json was collected with other kernel sources (sha {data.get('kernel_source_sha16')}, now {sha}): "
                      f"dropped as stale; refresh with tools/devcount/collect.py")
    rec = (data.get("configs", {}).get(config_name) or {}).get(path)
    return rec, (None if rec else f"profiles/devcount_traffic.json has no entry for {config_name} / {path}")


def roofline_of(path, n, counted_bytes_per_step, launch_us, traffic_rec, traffic_note, kernel, sha, moved_bytes_per_step=None):
    """The roofline object of one call shape on one leg.  ONE meaning per key, on every leg (VERDICT r4 "next" #2):
      frac          bytes the kernel moves BY CONSTRUCTION per launch (MOVED_BYTES: 38 CartPole, 34 where its reward store is elided, 18 MountainCar,
                    32.3 Pendulum) / launch time / peak -- = `achieved` / `peak`;
      frac_counted  the contract's algorithmic bytes (38 / 22 / 37) / launch time / peak: credits bytes an engine never stores;
      frac_moved    fabric bytes per launch seen by the device-wide counters (`traffic`, replayed from profiles/devcount_traffic.json: `traffic_from`)
                    / launch time / 8 TB/s; null without a figure for exactly these kernel sources.
    `peak` is the HBM roofline (8 TB/s) except for a cache-resident chain, whose state lives in the L2s between launches (`bound: "l2"`, 34.5 TB/s; no HBM
    fraction is claimed for it).  `hbm_bound` says whether the fabric really moves (>= 0.9 x) the bytes of `frac`: false = part of a step's arrays is served
    by the L2s / the Infinity Cache (`frac` is then a fraction of the HBM ROOFLINE, not HBM utilisation)."""
    moved = moved_bytes_per_step or counted_bytes_per_step
    seconds = launch_us * 1e-6
    achieved = n * moved / seconds / 1e9
    resident = (traffic_rec["bytes_per_launch"] < 0.9 * n * moved) if traffic_rec else (n * counted_bytes_per_step <= (128 << 20))
    in_l2 = path == "chain" and resident
    peak = L2_PEAK_GBPS if in_l2 else HBM_PEAK_GBPS
    roof = {"bound": "l2" if in_l2 else "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
            "frac_moved": None, "frac_counted": n * counted_bytes_per_step / seconds / 1e9 / peak,
            "traffic": None, "kernel": kernel, "bytes_per_env_step": moved, "bytes_per_env_step_counted": counted_bytes_per_step,
            "bytes_per_launch": n * moved, "launch_us": launch_us, "hbm_bound": (not resident) if traffic_rec else None, "kernel_source_sha16": sha}
    if traffic_rec:
        roof["traffic"] = traffic_rec["bytes_per_launch"]
        roof["traffic_from"] = "file:profiles/devcount_traffic.json"  # a replay of a counter run on these kernel sources, not a measurement of this run
        roof["traffic_read"] = traffic_rec.get("fetch_bytes")
        roof["traffic_written"] = traffic_rec.get("write_bytes")
        roof["traffic_over_bytes_by_construction"] = traffic_rec["bytes_per_launch"] / (n * moved)
        roof["frac_moved"] = traffic_rec["bytes_per_launch"] / seconds / 1e9 / HBM_PEAK_GBPS
    elif traffic_note:
        roof["traffic_note"] = traffic_note
    return roof


def moved_bytes(env_name: str, engine_extras: dict) -> float:
    """Bytes per env-step this engine moves by construction: CartPole engines of >= 128 MiB per step do not rewrite their constant reward (the engine says so)."""
    return MOVED_BYTES[env_name] - (4.0 if (env_name == "cartpole" and engine_extras.get("reward_store_elided")) else 0.0)


def cgroup_cpu_quota(root: str = "/sys/fs/cgroup"):
    """CPUs' worth of time the container's cgroup grants per period (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us); None = unlimited / unknown."""
    try:
        quota, period = (Path(root) / "cpu.max").read_text().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float((Path(root) / "cpu" / "cpu.cfs_quota_us").read_text())
        period = float((Path(root) / "cpu" / "cpu.cfs_period_us").read_text())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def usable_cpus(root: str = "/sys/fs/cgroup"):
    """(threads the CPU baseline's multi-thread leg runs, where that number came from)."""
    try:
        mask = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        mask = os.cpu_count() or 1
    quota = cgroup_cpu_quota(root)
    if quota is not None and quota < mask:
        return max(1, int(quota)), f"cgroup CPU quota {quota:g} (affinity mask {mask}, host {os.cpu_count()})"
    return max(1, mask), f"affinity mask {mask} CPUs, no smaller cgroup quota (host {os.cpu_count()})"


def cpu_baseline(kind: int, target_seconds: float):
    """The reference's single-env caller loop in the f64 C oracle, one host thread (kind "port":
    the reference is a Rust crate and cannot be built in this image)."""
    from oracle.bindings import Oracle

    orc = Oracle()
    secs, _ = orc.baseline_loop(kind, 2_000_000, 0, 0)  # calibrate
    rate = 2_000_000 / max(secs, 1e-9)
    n = int(min(max(rate * target_seconds, 2_000_000), 4e9))
    secs, out = orc.baseline_loop(kind, n, 0, 0)
    # also a multi-thread figure = one independent single-env loop per thread (the C
    # loop runs without the GIL).  Reported beside the single-thread value, never instead of it.
    import threading

    # As many threads as this process may really run at once ("all host cores, labelled as such" -- not a constant): the CPUs of
    # its affinity mask, capped by the container's cgroup CPU quota where there is one (the GPU boxes report 256 logical CPUs; 256 threads
    # under a quota of a few CPUs measured only 9x the single-thread rate).  The line says where the number came from.
    cores, cores_from = usable_cpus()
    per_thread = max(int(rate * min(target_seconds, 2.0)), 1_000_000)
    times = [0.0] * cores

    def work(i):
        times[i] = orc.baseline_loop(kind, per_thread, 0, i)[0]

    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    all_wall = time.perf_counter() - t0
    return {
        "value": n / secs,
        "unit": "env-steps/s",
        "cores": 1,
        "multi_thread": {"value": cores * per_thread / all_wall, "cores": cores,
                         "sample": f"{cores} independent single-env loops (threads) x {per_thread} steps, {all_wall:.1f} s wall, "
                                   f"host reports {os.cpu_count()} logical CPUs; threads = {cores_from}"},
        "kind": "port",
        "sample": f"{n} consecutive Env::step() calls of ONE env (f64 C restatement of gym-rs step()+reset, "
                  f"random actions, reset on done; loop shape of examples/cartpole.rs:15-30, RenderMode::None), "
                  f"{secs:.1f} s on 1 of {os.cpu_count()} host cores; mean episode length {out[1] / max(out[2], 1):.1f}",
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--env", choices=sorted(ENVS), default="cartpole")
    ap.add_argument("--n-envs", type=int, default=0, help="lanes per GPU (default: the BASELINE config)")
    ap.add_argument("--vec", type=int, default=0, help="lanes per work-item (4 or 8); 0 = engine default")
    ap.add_argument("--nt", type=int, default=0, help="memory hint: 0 auto, 1 always non-temporal, 2 never")
    ap.add_argument("--action-buffers", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline sample length; 0 disables it")
    ap.add_argument("--rollout", type=int, default=0, metavar="R",
                    help="NOT the headline: fused gymrs_rollout launches of R steps each (state stays in registers, "
                         "observations of intermediate steps are not materialised); reported with mode=fused_rollout")
    ap.add_argument("--record", action="store_true",
                    help="with --rollout: gymrs_rollout_record, i.e. every step's observation/action/reward/done is kept")
    ap.add_argument("--graph", action="store_true", help="replay captured HIP graphs (pays for small batches only)")
    ap.add_argument("--path", choices=["both", "per_step_visible", "chain"], default="both",
                    help="call shape(s) of the per-step API to time (docstring); the headline is per_step_visible unless only chain is asked for")
    ap.add_argument("--torch-allreduce", action="store_true",
                    help="sum the statistics with torch.distributed instead of the C ABI's own RCCL communicator")
    ap.add_argument("--native-rccl", action="store_true", help="(default since round 2; kept for old command lines)")
    ap.add_argument("--repetitions", type=int, default=REPETITIONS)
    ap.add_argument("--min-repetition-ms", type=float, default=MIN_REPETITION_SECONDS * 1e3,
                    help="a timed repetition lasts at least this long (default 100: the sustained rate, see MIN_REPETITION_SECONDS; runs under a profiler use 5)")
    ap.add_argument("--no-probe", action="store_true", help="skip the in-process copy-kernel probe (roofline.peak_measured)")
    ap.add_argument("--no-queue-shape", action="store_true", help="skip the extra measurement of the per-step-visible shape through the engine's queue (roofline.queue_launch_us)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short legs for the other BASELINE configs (MountainCar 2^20, Pendulum 2^22, CartPole 2^24) that "
                         "the default N=1 CartPole run attaches as `configs`")
    ap.add_argument("--pmc-traffic", action="store_true",
                    help="N=1: also run two short rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE) and "
                         "report the traffic they measure (and refresh profiles/pmc_traffic.json)")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process drives all N GPUs through the C ABI's native sharder (gymrs_sharded_*: one engine + one host thread per device, grouped RCCL "
                         "all-reduce) instead of one process per GPU; the line's `sharder` says which form ran.  With --oversubscribe the N blocks share the box's GPUs")
    ap.add_argument("--full-out", default="", help="where the complete record goes (default: bench_full.json next to this script); the printed line names it")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: ranks share GPUs (rank r -> device r %% device_count) and meet over gloo, so that the "
                         "N>1 code path (spawner, global env offsets, aggregation) runs on a 1-GPU box; not a benchmark")
    return ap.parse_args(argv)


class HipBackend:
    """The GPU side of one rank: engines from the C ABI, device action ring, HIP events on the engine's stream."""

    name = "hip"

    def __init__(self, args, info):
        import torch

        self.torch = torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no GPU visible; the stepper has no CPU fallback")
        n_dev = torch.cuda.device_count()
        self.oversubscribed = bool(args.oversubscribe and info.world > n_dev)
        self.dev_index = info.local_rank % n_dev if args.oversubscribe else info.local_rank
        if self.dev_index >= n_dev:
            raise SystemExit(f"bench.py: rank {info.rank} needs GPU {self.dev_index} but only {n_dev} are visible "
                             f"(one process per GPU)")
        torch.cuda.set_device(self.dev_index)
        self.device = torch.device("cuda", self.dev_index)
        # The control plane of a run (barrier, max over ranks, the agreement about which all-reduce to use) meets over gloo: it moves
        # a few doubles on the host and is the path the multi-rank tests exercise (CPU ranks, and 2 / 8 ranks sharing the one GPU of a
        # test box).  RCCL is used where the path has its one exchange, the statistics all-reduce, through the C ABI's own
        # communicator (sharded.setup_stats_allreduce: watched first contact, gloo as the fall-back on ALL ranks).  GYMRS_BENCH_NCCL=1
        # puts the control plane on torch.distributed's nccl (= RCCL) backend as well.
        self.collective_backend = "nccl" if (os.environ.get("GYMRS_BENCH_NCCL") == "1" and not self.oversubscribed) else "gloo"
        self.gymrs = importlib.import_module("gym-rs_amd")

    def make_engine(self, kind, n, offset, flags, vec):
        return self.gymrs.BatchedEngine(kind, n, global_env_offset=offset, device=self.dev_index, flags=flags,
                                        lanes_per_thread=vec or None)

    def probe_engine(self, kind, offset, flags):
        """A throw-away engine (own stream) for the first contact with RCCL: sharded.ShardedRun.setup_stats_allreduce."""
        return self.make_engine(kind, 256, offset, flags, 0)

    def make_action_ring(self, eng, n, nbuf, is_float):
        torch = self.torch
        ring = torch.empty((nbuf, n), dtype=torch.float32 if is_float else torch.uint8, device=self.device)
        torch.cuda.synchronize()
        for b in range(nbuf):
            eng.fill_actions(ring[b].data_ptr(), seed=1, t=b)
        return ring.data_ptr(), ring.stride(0) * ring.element_size(), ring

    def stream_of(self, eng):
        return self.torch.cuda.ExternalStream(eng.stream, device=self.device)

    def mark(self, stream):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    def sync(self):
        self.torch.cuda.synchronize()

    def copy_probe(self, read_bytes, write_bytes, launches, nt):
        """The copy yardstick: tools/copy_probe (a measurement tool, not part of the C ABI), loaded with ctypes."""
        if getattr(self, "_probe", None) is None:
            spec = importlib.util.spec_from_file_location("gymrs_copy_probe_tool", ROOT / "tools" / "copy_probe" / "build.py")
            self._probe = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(self._probe)
        return self._probe.copy_probe(self.dev_index, read_bytes, write_bytes, launches, nt)

    def trajectory_buffers(self, eng, n, rollout, is_float):
        torch = self.torch
        rs = (n + 15) // 16 * 16
        rec = {"obs": torch.empty((rollout, eng.obs_dim, rs), dtype=torch.float32, device=self.device),
               "actions": torch.empty((rollout, rs), dtype=torch.float32 if is_float else torch.uint8, device=self.device),
               "reward": torch.empty((rollout, rs), dtype=torch.float32, device=self.device),
               "done": torch.empty((rollout, rs), dtype=torch.uint8, device=self.device)}
        torch.cuda.synchronize()
        return rec


def choose_passes(seconds_per_pass: float, min_seconds: float = MIN_REPETITION_SECONDS) -> int:
    """Passes of K steps per timed repetition so that a repetition lasts at least min_seconds."""
    if seconds_per_pass <= 0:
        return 1
    return max(1, int(math.ceil(min_seconds / seconds_per_pass)))


def timed_repetitions(backend, coll, stream, run_pass, passes, repetitions):
    """R repetitions of `passes` x run_pass(); per repetition the MAX over ranks of (wall seconds, event ms)."""
    walls, kernels, own = [], [], []
    for _ in range(repetitions):
        coll.barrier()
        backend.sync()
        t0 = time.perf_counter()
        m0 = backend.mark(stream)
        for _ in range(passes):
            run_pass()
        m1 = backend.mark(stream)
        backend.sync()
        t1 = time.perf_counter()
        coll.barrier()
        own_ms = backend.elapsed_ms(m0, m1)
        wall, kms = coll.max([t1 - t0, own_ms])
        walls.append(wall)
        kernels.append(kms)
        own.append(own_ms)
    backend.own_event_ms = own  # this rank's own event times (the returned ones are the max over ranks)
    return walls, kernels


def load_pmc(name: str, env: str, sha: str):
    """A committed PMC figure, only if it was collected with the kernels that are running now."""
    path = ROOT / "profiles" / name
    try:
        data = json.loads(path.read_text())
    except Exception:
        return None, f"profiles/{name} missing"
    if data.get("kernel_source_sha16") != sha:
        return None, (f"profiles/{name} was collected with other kernel sources (sha {data.get('kernel_source_sha16')}, "
                      f"now {sha}): dropped as stale; refresh with bench.py --pmc-traffic")
    return data.get(env), None


def measure_pmc_traffic(args, env_name: str, sha: str):
    """PER-DISPATCH counters of this very command: two short rocprofv3 --pmc passes (one counter per pass, counters only), read back from the
    rocpd databases; FETCH_SIZE doubled as the guide's gfx950 correction for wide coalesced reads prescribes.  Both call shapes run in each
    pass (the HIP-launched kernel and the chain's copy of it are told apart by name).  rocprofv3's counter collection serialises kernels
    across queues, so the chains use the SYNCHRONOUS hand-over there (GYMRS_AQL_SYNC=1), and what the profiler does between two serialised
    kernels lies outside a kernel's counter window: these figures describe a kernel run IN ISOLATION (every launch finds the caches as the
    profiler left them), not the free-running launches the bench times -- those are in profiles/devcount_traffic.json."""
    import sqlite3
    import subprocess
    import tempfile

    out = {}
    note = None
    with tempfile.TemporaryDirectory(prefix="gymrs_pmc_", dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = Path(tmp) / ctr
            cmd = ["rocprofv3", "--pmc", ctr, "-d", str(d), "-o", "r", "--", sys.executable, str(ROOT / "bench.py"), "--env", env_name,
                   "--steps", "100", "--warmup", "20", "--cpu-seconds", "0", "--no-probe", "--no-configs", "--no-queue-shape", "--repetitions", "2", "--min-repetition-ms", "5"]
            if args.n_envs:
                cmd += ["--n-envs", str(args.n_envs)]
            if args.vec:
                cmd += ["--vec", str(args.vec)]
            env = dict(os.environ, TMPDIR="/tmp", GYMRS_AQL_SYNC="1")
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
            dbs = list(d.rglob("*_results.db"))
            if res.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {res.returncode}): {res.stderr[-300:]}"
            c = sqlite3.connect(str(dbs[0]))
            for path_name, glob in (("chain", f"gymrs_aql_{env_name}_f[0-9]_t*"), ("per_step_visible", "*step_kernel*")):  # (GLOB: `_` is a wildcard of LIKE)
                r = c.execute("select count(*), avg(value) from counters_collection where kernel_name glob ? and counter_name = ?", (glob, ctr)).fetchone()
                out.setdefault(path_name, {})[ctr] = {"launches": r[0], "avg_kb": r[1]}
    recs = {}
    for path_name, o in out.items():
        if not o["FETCH_SIZE"]["launches"] or not o["WRITE_SIZE"]["launches"]:
            continue  # (that call shape did not run under the profiler: e.g. no chains on this box)
        fetch = 2.0 * o["FETCH_SIZE"]["avg_kb"] * 1024.0
        write = o["WRITE_SIZE"]["avg_kb"] * 1024.0
        recs[path_name] = {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "raw": o}
    if not recs:
        return None, "rocprofv3 --pmc: no launch of the step kernel was counted"
    recs["correction"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests of wide coalesced reads as 64 B); WRITE_SIZE as reported"
    recs["what"] = "per-dispatch counters under rocprofv3's serialisation: each kernel in isolation, not free-running launches"
    path = ROOT / "profiles" / "pmc_traffic.json"
    try:
        data = json.loads(path.read_text())
        if data.get("kernel_source_sha16") != sha:
            data = {}
    except Exception:
        data = {}
    data["kernel_source_sha16"] = sha
    data[env_name] = recs
    try:
        path.write_text(json.dumps(data, indent=1))
    except Exception as exc:  # read-only checkout: the figure is still reported
        note = f"could not refresh profiles/pmc_traffic.json: {exc}"
    return recs, note


def run_rank(args, info, backend, make_collective=None):
    """Everything one rank does.  Returns the result dict on rank 0, None elsewhere.  `backend` supplies engines,
    action rings, timers (HipBackend here; tests/test_sharded_cpu.py passes a CPU stand-in built on the f32 twin,
    so that this very function -- sharding, repetitions, max-over-ranks, the statistics all-reduce, the line that is
    printed -- runs with world_size 2 over gloo)."""
    gymrs = importlib.import_module("gym-rs_amd")
    sharded = gymrs.sharded
    kind, n_default, bytes_read, bytes_written, workload = ENVS[args.env]
    bytes_per_step = bytes_read + bytes_written
    n = args.n_envs or n_default
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    if args.env == "pendulum":
        flags |= gymrs.TIME_LIMIT  # it never terminates: episodes end by the 200-step time limit only
    force_dist = os.environ.get("GYMRS_BENCH_FORCE_DIST") == "1"  # exercise the collective path with one rank
    coll = (make_collective or sharded.Collective)(info, backend.collective_backend, getattr(backend, "device", None), force=force_dist)
    run = sharded.ShardedRun(info, n, coll, lambda off, cnt: backend.make_engine(kind, cnt, off, flags, args.vec))
    eng = run.engine
    if args.nt:
        eng.set_tuning(args.vec or 4, args.nt)
    stream = backend.stream_of(eng)
    is_float = args.env == "pendulum"
    nbuf = max(1, args.action_buffers)
    act_ptr, act_stride, _ring = backend.make_action_ring(eng, n, nbuf, is_float)
    eng.reset(seed=0)
    rec = backend.trajectory_buffers(eng, n, args.rollout, is_float) if (args.rollout and args.record) else None

    steps_done = [0]  # steps since the last stats_clear (the run checks the all-reduced count against it)

    def run_steps(k):
        steps_done[0] += k
        if args.rollout:
            done = 0
            while done < k:
                r = min(args.rollout, k - done)
                if rec is not None:
                    eng.rollout_record(r, 1, done, obs=rec["obs"].data_ptr(), actions=rec["actions"].data_ptr(),
                                       reward=rec["reward"].data_ptr(), done=rec["done"].data_ptr(), lane_stride=rec["obs"].shape[2])
                else:
                    eng.rollout(r, action_seed=1, action_t0=done)
                done += r
        else:
            eng.step_many(act_ptr, act_stride, nbuf, k, use_graph=args.graph)

    # Which call shapes this run times (docstring): both unless --path names one; the fused rollout and graph replays are their own shape
    per_step = not (args.rollout or args.graph)
    paths = [HEADLINE_PATH, "chain"] if args.path == "both" else [args.path]
    if not per_step:
        paths = ["rollout" if args.rollout else "graph_replays"]
    head = paths[0]

    def extras():
        try:
            return json.loads(eng.env_json(0))["gymrs"] if hasattr(eng, "env_json") else {}
        except Exception:  # noqa: BLE001 -- diagnostics only
            return {}

    # ---- warm-up, communicator set-up, calibration: all outside the timed repetitions, on the headline's call shape ----
    with submission(head):
        run_steps(args.warmup)
        eng.sync()
        # (an oversubscribed TEST run cannot use RCCL at all: it refuses two ranks on one device)
        allreduce_path = run.setup_stats_allreduce(
            prefer_native=not (args.torch_allreduce or getattr(backend, "oversubscribed", False)),
            # first contact with RCCL happens on a throw-away engine with its own stream (sharded.setup_stats_allreduce says why)
            make_probe_engine=(lambda off: backend.probe_engine(kind, off, flags)) if hasattr(backend, "probe_engine") else None)
        if coll.active:
            run.allreduce_stats()  # RCCL builds its channels lazily on the first collective
        backend.sync()
        run_steps(args.steps)  # priming call
        backend.sync()
        t_settle = time.perf_counter()
        t0 = time.perf_counter()
        run_steps(args.steps)  # first calibration pass: clocks may still be ramping up after a short warm-up
        backend.sync()
        first = time.perf_counter() - t0
        pinned = os.environ.get("GYMRS_BENCH_PASSES")  # tests pin the amount of work to compare two runs' statistics
        # ~2 ms more for the rate, and in any case until the device has been stepping for SETTLE_SECONDS (see there)
        again = max(choose_passes(first, 2e-3), choose_passes(first, SETTLE_SECONDS) - 1)
        again = 1 if pinned else min(again, 4096)
        again = int(coll.max([again])[0])             # every rank steps the same number of times
        t0 = time.perf_counter()
        run_steps(args.steps * again)  # ONE call, like a timed repetition
        backend.sync()
        per_pass = (time.perf_counter() - t0) / again
        calibration_calls = [args.steps, args.steps, args.steps * again]
        # (the estimate came from a SHORT call, which overstates the time per step: top the settle phase up if it fell short)
        short = 0.0 if pinned else SETTLE_SECONDS - (time.perf_counter() - t_settle)
        more = int(coll.max([math.ceil(short / max(per_pass, 1e-9)) if short > 0 else 0])[0])
        if more > 0:
            run_steps(args.steps * more)
            backend.sync()
            calibration_calls.append(args.steps * more)
        passes = choose_passes(per_pass, args.min_repetition_ms * 1e-3)
        settle_ms = (time.perf_counter() - t_settle) * 1e3
        if pinned:
            passes = max(1, int(pinned))
        passes = int(coll.max([passes])[0])  # every rank times the same work
    eng.stats_clear()
    steps_done[0] = 0

    # ---- the timed region: R repetitions per call shape, the headline's first ----
    reps = max(1, args.repetitions)
    timed = {}
    for path in paths:
        with submission(path):
            before = extras().get("aql_launches", 0)
            if path != head:  # the other call shape: one untimed call of a repetition's length, so that its caches and clocks are its own
                run_steps(args.steps * passes)
                backend.sync()
            # One call per repetition: P passes of K steps = one gymrs_step_many(P * K).  One more repetition up front, reported but not
            # counted: whatever preceded the clock (stats_clear's kernels, the other call shape) has swept the caches.
            enqueue = []  # seconds until the call that enqueues a repetition returned: the launching thread's share

            def one_repetition():
                t = time.perf_counter()
                run_steps(args.steps * passes)
                enqueue.append(time.perf_counter() - t)

            walls, kernels = timed_repetitions(backend, coll, stream, one_repetition, 1, reps + 1)
            ex = extras()
            chained = ex.get("aql_launches", 0) - before
            timed[path] = {"walls": walls[1:], "kernels": kernels[1:], "lead_in_us": kernels[0] * 1e3 / (args.steps * passes),
                           "own_us": [ms * 1e3 / (args.steps * passes) for ms in getattr(backend, "own_event_ms", kernels)[-reps:]],
                           "enqueue_us": [e * 1e6 / (args.steps * passes) for e in enqueue[-reps:]],
                           "chained_launches": chained, "last_launch": ex.get("last_launch"), "handover": ex.get("aql_handover"),
                           "dispatcher": ex.get("aql")}

    # ---- read-out, after the clock: statistics all-reduce (the only collective of the path) ----
    backend.sync()
    t0 = time.perf_counter()
    total = run.allreduce_stats()
    stats_readout_us = (time.perf_counter() - t0) * 1e6
    run.check_total_steps(total, steps_done[0])

    def submission_of(path, t):
        if path not in PATHS:
            return None
        if path == "chain" and t["chained_launches"] <= 0:
            return "HIP launches -- no chain ran (the engine's dispatcher: %s)" % t["dispatcher"]
        if path == "chain":
            return "AQL chains: the engine's own HSA queue, one chain per gymrs_step_many call, agent-scope acquire on every launch, release fence only at the end of the chain (gymrs_aql.h)"
        return "HIP launches (hipLaunchKernelGGL): agent-scope acquire + release on every launch"

    mine = {"rank": info.rank, "device": getattr(backend, "dev_index", None), "global_env_offset": run.offset,
            "cpu_affinity": getattr(args, "cpu_affinity", None), "numa_node": getattr(args, "numa_node", None), "paths": {}}
    for path, t in timed.items():
        mine["paths"][path] = {"launch_us": statistics.median(t["own_us"]), "launch_us_min": min(t["own_us"]), "launch_us_max": max(t["own_us"]),
                               "submission": submission_of(path, t), "handover": t["handover"] if path == "chain" else None}
    hp = mine["paths"][head]  # the headline's figures also at the top of the rank record (what round 3's line carried)
    mine.update({"launch_us": hp["launch_us"], "launch_us_min": hp["launch_us_min"], "launch_us_max": hp["launch_us_max"], "submission": hp["submission"]})
    if coll.active:  # this rank's OWN statistics beside the all-reduced ones: when a sharded total is ever off, the record says which block and by how much
        try:
            mine["episodes"] = [float(x) for x in eng.stats()]
        except Exception as exc:  # noqa: BLE001 -- diagnostics only
            mine["episodes"] = repr(exc)
    per_rank = coll.gather_to_root(mine)
    out = None
    if info.is_root:
        sha = kernel_source_sha16()
        steps_timed = args.steps * passes
        config_name = {"cartpole": "cartpole_2p20", "mountain_car": "mountain_car_2p20", "pendulum": "pendulum_2p22"}[args.env] if (
            args.n_envs in (0, n_default) and not args.vec) else None
        path_out = {}
        for path, t in timed.items():
            wall = statistics.median(t["walls"])
            kernel_ms = statistics.median(t["kernels"])
            launch_us = kernel_ms * 1e3 / steps_timed
            prec = {"what": PATHS[path][1] if path in PATHS else path, "value": run.job_rate(steps_timed, wall), "unit": "env-steps/s",
                    "ms_per_step": wall * 1e3 / steps_timed, "launch_us": launch_us,
                    "event_us_per_step": {"min": min(t["kernels"]) * 1e3 / steps_timed, "median": launch_us, "max": max(t["kernels"]) * 1e3 / steps_timed,
                                          "spread": (max(t["kernels"]) - min(t["kernels"])) / kernel_ms},
                    "wall_ms_per_repetition": [w * 1e3 for w in t["walls"]], "event_ms_per_repetition": t["kernels"],
                    "lead_in_repetition_us_per_step": t["lead_in_us"], "submission": submission_of(path, t)}
            if t["enqueue_us"]:
                # rank 0's launching thread: how long the call that enqueues a repetition took, per launch.  Well below event_us_per_step = the device sets
                # the pace; close to it = the thread does (HIP launches cost it 3-4 us each, a chain's 0.9), and the figure is the host's, not the kernel's
                prec["host_enqueue_us_per_step"] = {"min": min(t["enqueue_us"]), "median": statistics.median(t["enqueue_us"]), "max": max(t["enqueue_us"])}
            if per_step:
                trec, tnote = load_free_running_traffic(config_name, path, sha) if config_name else (None, "no committed traffic figure for this size / tuning")
                prec["roofline"] = roofline_of(path, n, bytes_per_step, launch_us, trec, tnote, t["last_launch"], sha, moved_bytes(args.env, extras()))
                prec["roofline"]["how"] = ("HIP events on the engine's stream around ONE gymrs_step_many(P*K) call / (P*K), median of the repetitions: the back-to-back "
                                           "launches with their gaps (for a chain also its hand-over from and back to the stream); rank 0's lanes")
            path_out[path] = prec
        hd = path_out[head]
        out = {
            "metric": "env-steps/sec (whole node), CartPole-v1 @ 2^20 envs per MI355X" if args.env == "cartpole"
                      else f"env-steps/sec (whole node), {args.env}",
            "value": hd["value"],
            "unit": "env-steps/s",
            "n_gpus": info.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": hd["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "env": args.env,
                "lanes_per_gpu": n,
                "total_lanes": run.total_lanes,
                "flags": "|".join(nm for bit, nm in ((1, "AUTO_RESET"), (2, "TRACK_STATS"), (4, "TIME_LIMIT")) if flags & bit),
                "lanes_per_work_item": args.vec or 4,
                "action_buffers": nbuf,
                "hip_graph": bool(args.graph),
                "call_shape": head,
                "call_shape_note": PATHS[head][1] if head in PATHS else head,
                "submission": hd["submission"],
                "parallelism": f"lane-sharded x{info.world}, no data-path collective; 1 all-reduce of 4 f64 per run, after the clock",
                "stats_allreduce": allreduce_path,
                "control_plane": f"torch.distributed({backend.collective_backend}): barrier, max over ranks" if coll.active else "single process",
            },
            "timing": {
                "repetitions": reps,
                "passes_per_repetition": passes,
                # untimed calls between the warm-up and the first repetition, in steps per call (priming, rate, settle)
                "calibration_passes": sum(calibration_calls) // args.steps,
                "calibration_calls": calibration_calls,
                "lead_in_repetition_us_per_step": hd["lead_in_repetition_us_per_step"],  # the uncounted repetition right after stats_clear
                "settle_ms": settle_ms,  # untimed stepping right before the first repetition (calibration passes included)
                "steps_per_repetition": steps_timed,
                "wall_ms_per_repetition": hd["wall_ms_per_repetition"],
                "event_ms_per_repetition": hd["event_ms_per_repetition"],
                "event_us_per_step": hd["event_us_per_step"],
                "statistic": "median over repetitions of the max over ranks",
                "stats_readout_us": stats_readout_us,
                # CPUs this rank's launching thread and the HIP runtime's helpers were confined to (None = not pinned)
                "cpu_affinity": getattr(args, "cpu_affinity", None),
                "numa_node": getattr(args, "numa_node", None),  # the GPU's NUMA node when the block was taken from its local CPUs
            },
            "ranks": per_rank,
            "episodes": {"sum_return": float(total[0]), "sum_length": float(total[1]), "n_episodes": float(total[2])},
            "sharder": "process-per-gpu" if info.world > 1 else "single engine",
        }
        if per_step:
            out["roofline"] = hd["roofline"]
            out["paths"] = path_out
            # per-dispatch PMC figures next to the free-running ones: measured by this run (--pmc-traffic) or the committed file if it
            # belongs to these kernels
            pmc, pnote = (measure_pmc_traffic(args, args.env, sha) if (args.pmc_traffic and info.world == 1 and backend.name == "hip")
                          else load_pmc("pmc_traffic.json", args.env, sha))
            for path, prec in path_out.items():
                if pmc and pmc.get(path) and config_name:
                    prec["roofline"]["traffic_per_dispatch_pmc"] = {k: pmc[path][k] for k in ("bytes_per_launch", "fetch_bytes", "write_bytes")}
                    prec["roofline"]["traffic_per_dispatch_pmc"]["what"] = pmc.get("what", "rocprofv3 --pmc, per dispatch")
                elif pnote and args.pmc_traffic:
                    prec["roofline"]["traffic_per_dispatch_pmc_note"] = pnote
            if "chain" in path_out and head != "chain":
                out["paths"]["chain"]["reported_separately"] = ("not the headline: inside a chain nobody can read a step's observation / reward / done before "
                                                                "the chain ends (SURVEY H1: report multi-step variants separately)")
        # N > 1: do the ranks agree on how they submit?  job rate = max over ranks, so ONE rank on another submission path or hand-over costs the
        # whole job its difference, silently (VERDICT r3 "next" #8); and the rate the ranks' own launch times add up to, next to the measured one
        agree = True
        for path in timed:
            subs = {(r["paths"][path]["submission"], (r["paths"][path]["handover"] or "").split(" (")[0]) for r in per_rank}
            agree = agree and len(subs) == 1
        out["ranks_agree"] = agree
        if not agree:
            out["ranks_agree_note"] = "ranks differ in submission path or hand-over: see ranks[*].paths"
        out["expected_job_rate_from_rank_launch_times"] = sum(n / (r["launch_us"] * 1e-6) for r in per_rank)
        if run.allreduce_note:
            out["config"]["stats_allreduce_note"] = run.allreduce_note
        out["config"]["comm_watchdog"] = "a native RCCL call timed out and was abandoned" if run.abandoned else "not triggered"
        if getattr(backend, "oversubscribed", False):
            out["oversubscribed"] = "TEST RUN: ranks share GPUs and meet over gloo; value is not a benchmark result"
        if per_step:
            if not args.no_probe and info.world == 1 and hasattr(backend, "copy_probe"):  # (a one-GPU diagnostic: no rank keeps the others waiting)
                # the same box, the same process: what a plain copy gets (a) on HBM, (b) at this launch's footprint, submitted the way each
                # call shape is submitted: HIP launches (a release fence each) for per_step_visible, launches of a chain for chain -- like for like
                big = 1 << 30
                both = [u for u in (backend.copy_probe(big, big, 20, 1), backend.copy_probe(big, big, 20, 0)) if u]
                us_big = min(both) if both else None
                zeros = [u for u in (backend.copy_probe(big, big, 20, 16 | 1), backend.copy_probe(big, big, 20, 16)) if u]
                us_big_zeros = min(zeros) if zeros else None
                rd16, wr16 = n * bytes_read // 16 * 16, n * bytes_written // 16 * 16
                for path, prec in path_out.items():
                    base = 2 if path == "chain" else 0
                    cands = [u for u in (backend.copy_probe(rd16, wr16, 500, base | h) for h in (1, 0, 4, 9, 8, 12)) if u]
                    if not cands:
                        continue
                    us_same = min(cands)
                    roof = prec["roofline"]
                    roof["same_footprint_copy_us"] = us_same
                    zeros = [u for u in (backend.copy_probe(rd16, wr16, 500, 16 | base | h) for h in (1, 0, 4, 9, 8, 12)) if u]
                    if zeros:
                        roof["same_footprint_copy_of_zeros_us"] = min(zeros)  # (what every figure before evidence set r04e was: lines of zeros move faster)
                    roof["same_footprint_copy"] = (f"{bytes_read} B read + {bytes_written} B written per lane, {n} lanes per launch, back-to-back launches submitted like "
                                                   f"this call shape ({'launches of a chain' if path == 'chain' else 'HIP launches'}), in place like a step, a source of hashed 32-bit words (NOT zeros: profiles/r04_copy_content.log), the best of three hint choices x two shapes (4 / 1 items per work-item): "
                                                   "the floor of a step launch of this size")
                    roof["frac_of_same_footprint_copy"] = us_same / roof["launch_us"]
                    if us_big and path != "chain":
                        roof["peak_measured"] = {"hbm_copy_GBps": 2 * big / (us_big * 1e-6) / 1e9,
                                                 "hbm_copy": "1 GiB of hashed 32-bit words read + 1 GiB written per launch (beyond the 256 MiB Infinity Cache), dwordx4 copy kernel, one "
                                                             "item per work-item, the better of streaming-hinted and plain accesses"}
                        if us_big_zeros:
                            roof["peak_measured"]["hbm_copy_of_zeros_GBps"] = 2 * big / (us_big_zeros * 1e-6) / 1e9
                        roof["frac_of_measured_hbm_copy"] = roof["achieved"] / roof["peak_measured"]["hbm_copy_GBps"]
        else:
            # The fused kernel touches HBM once per launch of R steps: it is bound by VALU issue, not by HBM.  Its roofline is
            # the instruction issue rate: SQ_INSTS_VALU per launch (PMC, profiles/pmc_valu.json) / launch time against
            # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction.
            kernel_ms = statistics.median(timed[head]["kernels"])
            if args.rollout:
                n_launch = max(1, -(-args.steps // args.rollout)) * passes
                launch_us_r = kernel_ms * 1e3 / n_launch
                out["mode"] = "fused_rollout_recorded" if args.record else "fused_rollout"
                out["config"]["steps_per_launch"] = args.rollout
                out["config"]["workload"] = workload + f" -- fused rollout, {args.rollout} steps per launch (NOT the per-step headline)"
                key = args.env + ("_recorded" if args.record else "")
                rec_v, note = load_pmc("pmc_valu.json", key, sha)
                roof = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_WAVE_INSTR_PER_S / 1e9, "unit": "G wave-instr/s", "frac": None,
                        "traffic": None, "kernel": "rollout_kernel<%s, %d, flags=%d>" % (args.env, 4, flags), "launch_us": launch_us_r,
                        "ns_per_lane_step": kernel_ms * 1e6 / steps_timed / n, "kernel_source_sha16": sha}
                if rec_v and rec_v.get("steps_per_launch") == args.rollout and rec_v.get("lanes") == n:
                    per_launch = rec_v["SQ_INSTS_VALU_per_launch"]
                    roof["achieved"] = per_launch / (launch_us_r * 1e-6) / 1e9
                    roof["frac"] = roof["achieved"] / roof["peak"]
                    roof["valu_instr_per_wave_step"] = per_launch / (n / 256.0) / args.rollout
                    roof["how"] = ("SQ_INSTS_VALU per launch (profiles/pmc_valu.json, same kernel sources) / HIP-event launch time; peak = "
                                   "256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction (quarter-rate integer multiplies "
                                   "and LDS/branch issue slots make 1.0 unreachable)")
                elif note:
                    roof["note"] = note
                out["roofline"] = roof
            else:
                out["mode"] = "hip_graph_replays"
                launch_us = kernel_ms * 1e3 / steps_timed
                out["roofline"] = roofline_of("per_step_visible", n, bytes_per_step, launch_us, None, "graph replays: no committed traffic figure", extras().get("last_launch"), sha)
    if (info.is_root and info.world == 1 and backend.name == "hip" and per_step and HEADLINE_PATH in (out.get("paths") or {}) and hasattr(eng, "env_json")
            and not args.no_queue_shape):
        q_us = measure_visible_through_queue(backend, eng, stream, act_ptr, act_stride, nbuf, args.steps * max(1, passes // 5))
        if q_us:
            out["paths"][HEADLINE_PATH]["roofline"]["queue_launch_us"] = q_us  # (the same dict object as out["roofline"] when the headline is this shape)
    eng.close()
    if (info.is_root and info.world == 1 and backend.name == "hip" and per_step and args.path == "both" and not args.no_configs and args.env == "cartpole"
            and not args.n_envs and not args.vec and not args.nt):
        # the other BASELINE.json configs, driver-run: short legs after the headline (its engine is gone: one batch at a time)
        out["configs"] = {name: measure_config(backend, gymrs, name, *spec, no_probe=args.no_probe) for name, spec in EXTRA_CONFIGS.items()}
    if info.is_root:
        gymrs.sharded.restore_cpus(getattr(args, "cpu_affinity_before", None))  # the CPU baseline may use every core
        # N > 1: a SHORT sample (the other ranks wait at the barrier below), so that a scaling line still carries the baseline
        cpu_seconds = args.cpu_seconds if info.world == 1 else min(args.cpu_seconds, 2.0)
        if cpu_seconds > 0:
            out["cpu_baseline"] = (backend.cpu_baseline if hasattr(backend, "cpu_baseline") else cpu_baseline)(kind, cpu_seconds)
    coll.barrier()  # every rank leaves the process group together
    coll.close()
    args.hard_exit = run.abandoned  # a helper thread may still sit inside a native call that never returns
    return out


def measure_visible_through_queue(backend, eng, stream, act_ptr, act_stride, nbuf, steps_per_call, repetitions=5):
    """The per-step-visible shape submitted through the ENGINE'S OWN QUEUE instead of through HIP (GYMRS_AQL=2, VERDICT r4 "next" #4): every packet carries HIP's
    header (agent-scope acquire + release) and the launch uses HIP launches' memory hints, so every step's arrays are written back when its launch ends -- at
    0.3-0.5 us of host time per launch instead of 2.5-4.  Reported beside the HIP-launched figure (`queue_launch_us`), never instead of it: the faster of the two
    differs by env and size (profiles/r05_visible_through_queue.log).  None where the dispatcher is not available."""
    before = os.environ.get("GYMRS_AQL")
    os.environ["GYMRS_AQL"] = "2"
    try:
        launched = json.loads(eng.env_json(0))["gymrs"].get("aql_launches", 0)
        eng.step_many(act_ptr, act_stride, nbuf, steps_per_call)  # untimed: caches and clocks are this submission's own
        eng.sync()
        us = []
        for _ in range(repetitions):
            m0 = backend.mark(stream)
            eng.step_many(act_ptr, act_stride, nbuf, steps_per_call)
            m1 = backend.mark(stream)
            eng.sync()
            us.append(backend.elapsed_ms(m0, m1) * 1e3 / steps_per_call)
        if json.loads(eng.env_json(0))["gymrs"].get("aql_launches", 0) <= launched:
            return None
        return statistics.median(us)
    finally:
        if before is None:
            os.environ.pop("GYMRS_AQL", None)
        else:
            os.environ["GYMRS_AQL"] = before


def measure_config(backend, gymrs, config_name, env_name, n, nbuf, no_probe=False, repetitions=5):
    """One short leg for another BASELINE.json config on this GPU: same procedure as the headline (warm-up, settle, R repetitions of >= 5 ms
    of back-to-back launches between HIP events on the engine's stream, median), BOTH call shapes, reported as a sub-record whose top-level
    figures are the headline call shape's.  `roofline.frac` is what is MOVED wherever the fabric traffic of the leg is on file, next to
    `frac_counted` (the algorithmic bytes the contract counts: MountainCar's elided constant reward store and Pendulum's aliased theta_dot
    column are counted there and not moved; VERDICT r3 "next" #1e)."""
    kind, _, bytes_read, bytes_written, workload = ENVS[env_name]
    bytes_per_step = bytes_read + bytes_written
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | (gymrs.TIME_LIMIT if env_name == "pendulum" else 0)
    eng = backend.make_engine(kind, n, 0, flags, 0)
    stream = backend.stream_of(eng)
    act_ptr, act_stride, _ring = backend.make_action_ring(eng, n, nbuf, env_name == "pendulum")
    eng.reset(seed=0)
    k = 64
    sha = kernel_source_sha16()

    def timed(n_pass):
        m0 = backend.mark(stream)
        for _ in range(n_pass):
            eng.step_many(act_ptr, act_stride, nbuf, k)
        m1 = backend.mark(stream)
        eng.sync()
        return backend.elapsed_ms(m0, m1)

    rec = {"workload": (workload if n == ENVS[env_name][1] else f"{env_name} @ {n} envs") + f", ring of {nbuf} action buffers", "lanes": n, "action_buffers": nbuf,
           "bytes_per_env_step": bytes_per_step, "repetitions": repetitions, "paths": {}}
    for path in (HEADLINE_PATH, "chain"):
        with submission(path):
            before = json.loads(eng.env_json(0))["gymrs"].get("aql_launches", 0)
            timed(2)
            per_pass = max(timed(4) / 4, 1e-3)                                   # ms per 64 steps
            timed(max(1, int(SETTLE_SECONDS * 1e3 / per_pass)))                  # settle (untimed)
            n_pass = max(1, int(math.ceil(MIN_REPETITION_SECONDS * 1e3 / per_pass)))
            us = sorted(timed(n_pass) * 1e3 / (n_pass * k) for _ in range(repetitions))
            ex = json.loads(eng.env_json(0))["gymrs"]
        launch_us = statistics.median(us)
        trec, tnote = load_free_running_traffic(config_name, path, sha)
        roof = roofline_of(path, n, bytes_per_step, launch_us, trec, tnote, ex.get("last_launch"), sha, moved_bytes(env_name, ex))
        prec = {"value": n / (launch_us * 1e-6), "unit": "env-steps/s", "launch_us": launch_us, "launch_us_min": us[0], "launch_us_max": us[-1],
                "steps_per_repetition": n_pass * k, "roofline": roof,
                "submission": ("chain" if ex.get("aql_launches", 0) > before else "HIP launches") if path == "chain" else "HIP launches"}
        elided = 4 if (env_name == "cartpole" and ex.get("reward_store_elided")) else 0  # (the copy floor then copies 17 + 17)
        roof["reward_store_elided"] = bool(elided)
        if not no_probe:
            rd16, wr16 = n * bytes_read // 16 * 16, n * (bytes_written - elided) // 16 * 16
            base = 2 if path == "chain" else 0
            cands = [u for u in (backend.copy_probe(rd16, wr16, max(20, int(2e3 / launch_us)), base | h) for h in (1, 0, 4, 9, 8, 12)) if u]
            if cands:
                roof["same_footprint_copy_us"] = min(cands)
                roof["frac_of_same_footprint_copy"] = min(cands) / launch_us
        rec["paths"][path] = prec
    q_us = measure_visible_through_queue(backend, eng, stream, act_ptr, act_stride, nbuf, k * max(1, int(math.ceil(20.0 / max(per_pass, 1e-3)))))
    if q_us:
        rec["paths"][HEADLINE_PATH]["roofline"]["queue_launch_us"] = q_us
    stats = eng.stats()
    eng.close()
    del _ring
    hd = rec["paths"][HEADLINE_PATH]
    rec.update({"value": hd["value"], "unit": "env-steps/s", "launch_us": hd["launch_us"], "launch_us_min": hd["launch_us_min"], "launch_us_max": hd["launch_us_max"],
                "roofline": hd["roofline"], "call_shape": HEADLINE_PATH, "episodes_finished": float(stats[2])})
    return rec


LINE_LIMIT = 4096  # bytes: the driver keeps the last 8 KB of stdout; round 4's 35 KB line came back as `parsed: null`


def _sig(x, digits=6):
    """Numbers of the printed line with 6 significant digits (the full record keeps every digit)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _roof_brief(r, keys=("bound", "achieved", "peak", "unit", "frac", "frac_moved", "frac_counted", "traffic", "kernel", "bytes_per_launch", "launch_us",
                         "hbm_bound", "traffic_from", "frac_of_same_footprint_copy", "queue_launch_us")):
    return {k: (r[k][:120] if isinstance(r[k], str) else _sig(r[k])) for k in keys if k in r}


def compact_line(out: dict, full_path) -> str:
    """THE line (the last line of stdout): the contract's fields, the headline's roofline, the CPU baseline and one small record per other
    call shape / BASELINE config -- numbers and names only, no prose.  Everything else (per-repetition times, per-rank records, notes, probes)
    is in the full record `full` names (and on stderr)."""
    line = {k: _sig(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                      "dtype", "data") if k in out}
    c = out.get("config", {})
    line["config"] = {k: (c[k][:200] if isinstance(c[k], str) else c[k])
                      for k in ("workload", "call_shape", "lanes_per_gpu", "total_lanes", "parallelism", "stats_allreduce", "steps_per_launch") if k in c}
    if out.get("mode"):
        line["mode"] = out["mode"]
    t = out.get("timing", {})
    if t:
        line["timing"] = {"repetitions": t.get("repetitions"), "steps_per_repetition": t.get("steps_per_repetition"),
                          "event_us_per_step": {k: _sig(v) for k, v in (t.get("event_us_per_step") or {}).items()}}
    if "roofline" in out:
        line["roofline"] = _roof_brief(out["roofline"], ("bound", "achieved", "peak", "unit", "frac", "frac_moved", "frac_counted", "traffic", "kernel", "bytes_per_launch",
                                                         "launch_us", "hbm_bound", "traffic_from", "frac_of_same_footprint_copy", "queue_launch_us", "valu_instr_per_wave_step", "ns_per_lane_step"))
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb.get("sample_brief") or cb.get("sample", "")[:160]}
        if cb.get("multi_thread"):
            line["cpu_baseline"]["multi_thread"] = {"value": _sig(cb["multi_thread"]["value"]), "cores": cb["multi_thread"]["cores"]}
    small = ("bound", "frac", "frac_moved", "frac_counted", "queue_launch_us")
    for name, p in (out.get("paths") or {}).items():
        if name != c.get("call_shape"):
            line.setdefault("paths", {})[name] = {"value": _sig(p["value"]), "launch_us": _sig(p["launch_us"]), **_roof_brief(p.get("roofline", {}), small)}
    for name, cf in (out.get("configs") or {}).items():
        line.setdefault("configs", {})[name] = {"value": _sig(cf["value"]), "launch_us": _sig(cf["launch_us"]), **_roof_brief(cf.get("roofline", {}), small)}
    if "ranks" in out and out.get("n_gpus", 1) > 1:
        line["ranks_launch_us"] = [_sig(r.get("launch_us"), 4) for r in out["ranks"]]
        line["ranks_agree"] = out.get("ranks_agree")
    for k in ("sharder", "oversubscribed"):
        if k in out:
            line[k] = out[k] if k == "sharder" else True
    line["full"] = str(full_path) if full_path else None
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:  # never again a line the driver cannot keep: shed the optional blocks, largest first
        for k in ("configs", "paths", "ranks_launch_us", "timing"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return text


def write_full_record(out: dict, where) -> "Path | None":
    """The complete record: to the file --full-out names (default bench_full.json next to this script; /tmp when the checkout is read-only)
    and, pretty-printed, to stderr.  Returns the path that was written."""
    text = json.dumps(out, indent=1)
    sys.stderr.write("bench.py full record:\n" + text + "\n")
    sys.stderr.flush()
    for cand in ([Path(where)] if where else [ROOT / "bench_full.json", Path("/tmp") / f"gymrs_bench_full_{os.getpid()}.json"]):
        try:
            cand.write_text(text + "\n")
            return cand
        except OSError:
            continue
    return None


def run_in_process(args):
    """`--in-process`: the whole job in ONE process through the C ABI's native sharder (include/gymrs_amd.h gymrs_sharded_*): a batch
    of N x lanes_per_gpu lanes cut into N contiguous blocks, one engine and one native host thread per device, the statistics summed by ONE grouped RCCL
    all-reduce after the clock (host-side sum where blocks share a device).  Same workload, same timing rules as the process-per-GPU form: W warm-up steps, a
    settle phase, R repetitions of one gymrs_sharded_step_many(P * K) each between synchronises (wall clock) and between HIP events on EVERY block's stream
    (the repetition's event time = the max over blocks), the median over repetitions."""
    import torch

    gymrs = importlib.import_module("gym-rs_amd")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the stepper has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.gpus > n_dev and not args.oversubscribe:
        raise SystemExit(f"bench.py --in-process: --gpus {args.gpus} but only {n_dev} GPUs are visible (one block per GPU; --oversubscribe shares them: TEST only)")
    devices = [r % n_dev for r in range(args.gpus)]
    kind, n_default, bytes_read, bytes_written, workload = ENVS[args.env]
    n = args.n_envs or n_default
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | (gymrs.TIME_LIMIT if args.env == "pendulum" else 0)
    sh = gymrs.ShardedEngine(kind, n * args.gpus, devices, flags=flags)
    is_float = args.env == "pendulum"
    esz = 4 if is_float else 1
    nbuf = max(1, args.action_buffers)
    pitch = max(s.n_envs for s in sh.shards)
    rings = [torch.empty((nbuf, pitch), dtype=torch.float32 if is_float else torch.uint8, device=f"cuda:{s.device}") for s in sh.shards]
    for b in range(nbuf):
        sh.fill_actions([r[b].data_ptr() for r in rings], seed=1, t=b)
    sh.reset(seed=0)
    ptrs = [r.data_ptr() for r in rings]
    streams = [torch.cuda.ExternalStream(s.stream, device=torch.device("cuda", s.device)) for s in sh.shards]

    def run_steps(k):
        sh.step_many(ptrs, pitch * esz, nbuf, k)

    def timed(k):
        sh.sync()
        marks = []
        for st, s in zip(streams, sh.shards):
            with torch.cuda.device(s.device):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(st)
                marks.append(ev)
        t0 = time.perf_counter()
        run_steps(k)
        ends = []
        for st, s in zip(streams, sh.shards):
            with torch.cuda.device(s.device):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(st)
                ends.append(ev)
        sh.sync()
        wall = time.perf_counter() - t0
        return wall, [a.elapsed_time(b) for a, b in zip(marks, ends)]

    with submission(HEADLINE_PATH if args.path in ("both", HEADLINE_PATH) else "chain"):
        head = HEADLINE_PATH if args.path in ("both", HEADLINE_PATH) else "chain"
        run_steps(args.warmup)
        sh.sync()
        first, _ = timed(args.steps)
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < SETTLE_SECONDS:
            run_steps(max(args.steps, int(2e-3 / max(first / args.steps, 1e-9))))
            sh.sync()
        per_pass, _ = timed(args.steps * 8)
        passes = int(os.environ.get("GYMRS_BENCH_PASSES") or choose_passes(per_pass / 8, args.min_repetition_ms * 1e-3))
        sh.stats_clear()
        reps = max(1, args.repetitions)
        walls, events = [], []
        for _ in range(reps + 1):  # (one uncounted lead-in repetition, as in the process-per-GPU form)
            w, ev = timed(args.steps * passes)
            walls.append(w)
            events.append(ev)
        walls, events = walls[1:], events[1:]
    total = sh.stats()
    reduce_path = sh.reduce_path
    steps_timed = args.steps * passes
    assert total[3] == float(n * args.gpus) * steps_timed * (reps + 1), (total, steps_timed)
    wall = statistics.median(walls)
    per_block_us = [statistics.median(e[r] for e in events) * 1e3 / steps_timed for r in range(args.gpus)]
    launch_us = statistics.median(max(e) for e in events) * 1e3 / steps_timed
    sha = kernel_source_sha16()
    ex = json.loads(sh.shards[0].env_json(0))["gymrs"]
    config_name = "cartpole_2p20" if (args.env == "cartpole" and n == n_default and args.gpus == 1) else None
    trec, tnote = load_free_running_traffic(config_name, head, sha) if config_name else (None, "no committed traffic figure for this shape")
    roof = roofline_of(head, n, bytes_read + bytes_written, per_block_us[0], trec, tnote, ex.get("last_launch"), sha, moved_bytes(args.env, ex))
    ev_us = sorted(max(e) * 1e3 / steps_timed for e in events)
    out = {
        "metric": "env-steps/sec (whole node), CartPole-v1 @ 2^20 envs per MI355X" if args.env == "cartpole" else f"env-steps/sec (whole node), {args.env}",
        "value": n * args.gpus * steps_timed / wall, "unit": "env-steps/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / steps_timed, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "env": args.env, "lanes_per_gpu": n, "total_lanes": n * args.gpus, "call_shape": head, "action_buffers": nbuf,
                   "parallelism": f"lane-sharded x{args.gpus} in one process, no data-path collective; 1 all-reduce of 4 f64 per run, after the clock",
                   "stats_allreduce": {"rccl": "gymrs_sharded_stats: grouped RCCL all-reduce (one communicator over the blocks' engines)",
                                       "host": "gymrs_sharded_stats: host-side sum (blocks share a device, or one block)"}.get(reduce_path, reduce_path),
                   "devices": devices},
        "timing": {"repetitions": reps, "passes_per_repetition": passes, "steps_per_repetition": steps_timed, "wall_ms_per_repetition": [w * 1e3 for w in walls],
                   "event_us_per_step": {"min": ev_us[0], "median": statistics.median(ev_us), "max": ev_us[-1], "spread": (ev_us[-1] - ev_us[0]) / statistics.median(ev_us)},
                   "statistic": "median over repetitions of the max over blocks"},
        "roofline": roof,
        "ranks": [{"rank": r, "device": s.device, "global_env_offset": s.global_env_offset, "launch_us": per_block_us[r]} for r, s in enumerate(sh.shards)],
        "ranks_agree": True,
        "episodes": {"sum_return": float(total[0]), "sum_length": float(total[1]), "n_episodes": float(total[2])},
        "sharder": "in-process",
        "sharder_note": "gymrs_sharded_* (include/gymrs_amd.h): one engine + one native host thread per block; bit-identical to one engine of the same lanes "
                        "(tests/test_gpu_sharded_native.py)",
    }
    if args.oversubscribe and args.gpus > n_dev:
        out["oversubscribed"] = "TEST RUN: blocks share GPUs; value is not a benchmark result"
    del launch_us
    if args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(kind, min(args.cpu_seconds, 2.0) if args.gpus > 1 else args.cpu_seconds)
    sh.close()
    return out


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    gymrs = importlib.import_module("gym-rs_amd")
    sharded = gymrs.sharded
    if args.in_process:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)  # (RCCL's banner and friends: to stderr, as in the process-per-GPU form)
        out = run_in_process(args)
        os.write(json_fd, (compact_line(out, write_full_record(out, args.full_out)) + "\n").encode())
        os.close(json_fd)
        return 0
    if sharded.needs_spawn(args.gpus, force=os.environ.get("GYMRS_BENCH_FORCE_SPAWN") == "1"):
        # plain `python bench.py --gpus N`: become the launcher of N ranks, one process per GPU
        return sharded.spawn_ranks(str(Path(__file__).resolve()), argv, args.gpus)
    info = sharded.rank_info()
    if info.world != args.gpus:
        if info.is_root:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={info.world} ranks", file=sys.stderr)
        return 2
    # stdout carries ONE line, rank 0's JSON: whatever else writes to file descriptor 1 in a rank process (RCCL's version
    # banner -- flushed at exit, i.e. AFTER the JSON line --, gloo's connection chatter) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # a few CPUs per rank for the launching thread and the HIP runtime's helpers (sharded.pin_rank_to_cpus says why);
    # the CPU baseline leg gets the full mask back
    # -- taken from the CPUs local to the rank's GPU (its NUMA node) when the topology can be read
    args.cpu_affinity, args.cpu_affinity_before, args.numa_node = sharded.pin_rank_near_gpu(info.local_rank, n_local_ranks=info.world)
    backend = HipBackend(args, info)
    try:
        out = run_rank(args, info, backend)
    except TimeoutError as exc:  # a native collective that never completes (sharded.ShardedRun.allreduce_stats): its helper thread cannot be joined
        sys.stderr.write(f"bench.py: {exc}\n")
        sys.stderr.flush()
        os._exit(3)
    if out is not None:
        full_path = write_full_record(out, args.full_out)
        os.write(json_fd, (compact_line(out, full_path) + "\n").encode())
    os.close(json_fd)
    if getattr(args, "hard_exit", False):
        sys.stderr.flush()
        os._exit(0)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
