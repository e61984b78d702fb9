#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched stepper (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE Env::step() of every lane of the workload = one launch of the step kernel over the
rank's shard (auto-reset and statistics on).  Workload at N=1: BASELINE.json configs[1], CartPole-v1
at 2^20 parallel envs, f32.  At N>1 every rank holds 2^20 lanes (weak scaling; configs[4] at N=8 is
2^23 lanes) with global env ids rank*2^20+i and no data-path collective; the only collective is one
RCCL all-reduce of the 4 statistics doubles at the end of the timed region.

Inputs are resident in HBM before the timed region: the state arrays, and a ring of pre-generated
random-policy action buffers (the `rng.gen_range(0..=1)` of examples/cartpole.rs:19, produced on the
device by the Philox action stream).

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how each field is obtained.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENVS = {
    # name: (kind, default lanes per GPU, algorithmic bytes per env-step, workload label)
    "cartpole": (0, 1 << 20, 38, "CartPole-v1 @ 2^20 envs per GPU, f32, auto-reset, random policy"),
    "mountain_car": (1, 1 << 20, 22, "MountainCar-v0 @ 2^20 envs per GPU, f32, auto-reset, random policy"),
    "pendulum": (2, 1 << 22, 37, "Pendulum-v1 (spec-derived) @ 2^22 envs per GPU, f32, auto-reset, 200-step time limit, random policy"),
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


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

    # 16 threads at most: the GPU boxes report 256 logical CPUs but a container CPU quota of far fewer
    # (256 threads measured only 9x the single-thread rate), and the sample has to stay short.
    cores = min(os.cpu_count() or 1, 16)
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
                                   f"host reports {os.cpu_count()} logical CPUs"},
        "kind": "port",
        "sample": f"{n} consecutive Env::step() calls of ONE env (f64 C restatement of gym-rs step()+reset, "
                  f"random actions, reset on done; loop shape of examples/cartpole.rs:15-30, RenderMode::None), "
                  f"{secs:.1f} s on 1 of {os.cpu_count()} host cores; mean episode length {out[1] / max(out[2], 1):.1f}",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--env", choices=sorted(ENVS), default="cartpole")
    ap.add_argument("--n-envs", type=int, default=0, help="lanes per GPU (default: the BASELINE config)")
    ap.add_argument("--vec", type=int, default=0, help="lanes per work-item (4, 8, 16); 0 = engine default")
    ap.add_argument("--nt", type=int, default=0, help="memory hint: 0 auto, 1 always non-temporal, 2 never")
    ap.add_argument("--action-buffers", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline sample length; 0 disables it")
    ap.add_argument("--rollout", type=int, default=0, metavar="R",
                    help="NOT the headline: fused gymrs_rollout launches of R steps each (state stays in registers, "
                         "observations of intermediate steps are not materialised); reported with mode=fused_rollout")
    ap.add_argument("--record", action="store_true",
                    help="with --rollout: gymrs_rollout_record, i.e. every step's observation/action/reward/done is kept")
    ap.add_argument("--graph", action="store_true", help="replay captured HIP graphs (pays for small batches only)")
    ap.add_argument("--native-rccl", action="store_true", help="all-reduce through the C ABI's RCCL path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the stepper has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("GYMRS_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank
    dist_on = world > 1 or (force_dist and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    gymrs = importlib.import_module("gym-rs_amd")
    kind, n_default, bytes_per_step, workload = ENVS[args.env]
    n = args.n_envs or n_default
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    if args.env == "pendulum":
        flags |= gymrs.TIME_LIMIT  # it never terminates: episodes end by the 200-step time limit only
    eng = gymrs.BatchedEngine(kind, n, global_env_offset=rank * n, device=local_rank, flags=flags,
                              lanes_per_thread=args.vec or None)
    if args.nt:
        eng.set_tuning(args.vec or 4, args.nt)
    stream = torch.cuda.ExternalStream(eng.stream, device=local_rank)

    # synthetic inputs, resident in HBM before the timed region
    act_dtype = torch.float32 if args.env == "pendulum" else torch.uint8
    nbuf = max(1, args.action_buffers)
    actions = torch.empty((nbuf, n), dtype=act_dtype, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    for b in range(nbuf):
        eng.fill_actions(actions[b].data_ptr(), seed=1, t=b)
    stride = actions.stride(0) * actions.element_size()
    eng.reset(seed=0)

    rec = None
    if args.rollout and args.record:
        rs = (n + 15) // 16 * 16
        dev = f"cuda:{local_rank}"
        rec = {"obs": torch.empty((args.rollout, eng.obs_dim, rs), dtype=torch.float32, device=dev),
               "actions": torch.empty((args.rollout, rs), dtype=act_dtype, device=dev),
               "reward": torch.empty((args.rollout, rs), dtype=torch.float32, device=dev),
               "done": torch.empty((args.rollout, rs), dtype=torch.uint8, device=dev)}
        torch.cuda.synchronize()

    def run(k):
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
            eng.step_many(actions.data_ptr(), stride, nbuf, k, use_graph=args.graph)

    run(args.warmup)
    eng.sync()
    eng.stats_clear()
    if args.native_rccl and dist_on:
        uid = [eng.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(world, rank, uid[0])

    if dist_on and not args.native_rccl:
        # RCCL builds its communicator lazily on the first collective: do one outside the timed region
        dist.all_reduce(torch.zeros(4, dtype=torch.float64, device=f"cuda:{local_rank}"))
    elif dist_on:
        eng.allreduce_stats()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    run(args.steps)
    ev1.record(stream)
    if dist_on:
        if args.native_rccl:
            total = eng.allreduce_stats()
        else:
            st = torch.tensor(eng.stats(), dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(st)  # RCCL over xGMI: 32 bytes per rank
            total = st.cpu().numpy()
    else:
        total = eng.stats()
    eng.sync()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t1 = time.perf_counter()

    wall = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1)  # HIP events on the engine's stream around the K launches
    if dist_on:
        tmax = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(tmax[0]), float(tmax[1])
    total_steps = float(total[3])
    assert total_steps == float(n) * args.steps * world, (total_steps, n, args.steps, world)

    if rank == 0:
        value = total_steps / wall
        launch_us = kernel_ms * 1e3 / args.steps
        achieved = n * bytes_per_step / (launch_us * 1e-6) / 1e9
        traffic = None
        tr_file = ROOT / "profiles" / "pmc_traffic.json"
        if tr_file.exists():
            try:
                traffic = json.loads(tr_file.read_text()).get(args.env, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node), CartPole-v1 @ 2^20 envs per MI355X" if args.env == "cartpole"
                      else f"env-steps/sec (whole node), {args.env}",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "env": args.env,
                "lanes_per_gpu": n,
                "total_lanes": n * world,
                "flags": "|".join(nm for bit, nm in ((1, "AUTO_RESET"), (2, "TRACK_STATS"), (4, "TIME_LIMIT")) if flags & bit),
                "lanes_per_work_item": args.vec or 4,
                "action_buffers": nbuf,
                "hip_graph": bool(args.graph),
                "parallelism": f"lane-sharded x{world}, no data-path collective; 1 RCCL all-reduce of 4 f64 per run",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "kernel": "step_kernel<%s, 4, flags=%d>" % (args.env, flags),
                "bytes_per_env_step": bytes_per_step,
                "bytes_per_launch": n * bytes_per_step,
                "launch_us": launch_us,
                "how": "HIP events on the engine stream around the K timed launches / K (includes inter-kernel gaps)",
            },
            "episodes": {"sum_return": float(total[0]), "sum_length": float(total[1]), "n_episodes": float(total[2])},
        }
        if args.rollout:
            # the fused kernel touches HBM once per launch of R steps: it is VALU-bound, the HBM roofline of the
            # per-step kernel does not apply and is not claimed
            out["mode"] = "fused_rollout_recorded" if args.record else "fused_rollout"
            out["config"]["steps_per_launch"] = args.rollout
            out["config"]["workload"] = workload + f" -- fused rollout, {args.rollout} steps per launch (NOT the per-step headline)"
            out["roofline"] = {"bound": "valu", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                               "kernel": "rollout_kernel<%s, 4, flags=%d>" % (args.env, flags),
                               "launch_us": kernel_ms * 1e3 / max(1, -(-args.steps // args.rollout)),
                               "ns_per_lane_step": kernel_ms * 1e6 / args.steps / n}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(kind, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist_on:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
