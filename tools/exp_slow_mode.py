#!/usr/bin/env python
"""Developer experiment (VERDICT r2 "next" #2): what is the slow mode (6.4 -> 6.9 us per 2^20-lane CartPole step)?

One pinned process steps the headline engine in repetitions of 740 launches (the driver form's repetition) for
--seconds and records, per repetition: HIP-event us per step (eager), and every --probe-every repetitions also
  * the same through graph replays (host out of the picture),
  * the in-library copy probe at the step's footprint (memory system only, no VALU work),
  * a chip-wide VALU-only kernel (time ~ 1 / sclk) and the shader clock measured ON the device while the steps run
    (tools/clock_probe.hip: s_memtime against the 100 MHz s_memrealtime),
while a sampler thread (optional, --smi-ms) reads amdsmi's gpu_metrics (gfx / memory / fabric clocks, socket power,
throttle status, temperatures).  Prints one JSON object: the repetitions, the samples, and a summary that lines the slow
repetitions up against the fast ones for every quantity measured.

    python tools/exp_slow_mode.py [--seconds 8] [--smi-ms 5] [--probe-every 4] [--tag NAME]
"""
import argparse
import ctypes as C
import importlib
import json
import os
import statistics
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def smi_open():
    try:
        import amdsmi

        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        return amdsmi, hs[0]
    except Exception as exc:  # noqa: BLE001
        return None, repr(exc)


def smi_sample(amdsmi, h):
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    keep = {}
    for k in ("current_gfxclk", "current_uclk", "current_socclk", "average_gfxclk_frequency", "average_uclk_frequency",
              "average_socclk_frequency", "average_fclk_frequency", "average_socket_power", "current_socket_power", "throttle_status",
              "indep_throttle_status", "temperature_hotspot", "temperature_mem", "average_gfx_activity", "average_umc_activity",
              "gfx_activity_acc", "mem_activity_acc", "pcie_bandwidth_inst", "accumulation_counter", "prochot_residency_acc",
              "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "firmware_timestamp",
              "system_clock_counter"):
        v = m.get(k)
        if isinstance(v, (int, float)):
            keep[k] = v
    g = m.get("current_gfxclks")
    if isinstance(g, (list, tuple)):
        vals = [x for x in g if isinstance(x, (int, float)) and 0 < x < 60000]
        if vals:
            keep["gfxclks_min"], keep["gfxclks_max"], keep["gfxclks_mean"] = min(vals), max(vals), sum(vals) / len(vals)
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--steps", type=int, default=740)
    ap.add_argument("--smi-ms", type=float, default=5.0, help="0 = no sampler thread")
    ap.add_argument("--probe-every", type=int, default=4, help="0 = eager repetitions only")
    ap.add_argument("--warm-steps", type=int, default=50, help="the driver form warms up with 5 + 20 + ~300 steps only")
    ap.add_argument("--tag", default="")
    ap.add_argument("--no-pin", action="store_true")
    args = ap.parse_args()

    gymrs = importlib.import_module("gym-rs_amd")
    affinity = None
    if not args.no_pin:
        affinity, _ = gymrs.sharded.pin_rank_to_cpus(0)
    import torch

    lib = gymrs.load_library()
    probe = C.CDLL(str(ROOT / "tools" / "libclockprobe.so"))
    probe.clock_probe.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    probe.alu_probe.argtypes = [C.c_int, C.POINTER(C.c_double)]

    n, nbuf = 1 << 20, 32
    ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    eng = gymrs.BatchedEngine(0, n, flags=3)
    eng.reset(seed=0)
    for b in range(nbuf):
        eng.fill_actions(ring[b].data_ptr(), seed=1, t=b)
    stream = torch.cuda.ExternalStream(eng.stream, device=torch.device("cuda", 0))
    eng.step_many(ring.data_ptr(), n, nbuf, args.warm_steps)
    eng.sync()

    samples, stop = [], threading.Event()
    amdsmi, h = (None, "off")
    if args.smi_ms > 0:
        amdsmi, h = smi_open()
    t_start = time.perf_counter()

    def sampler():
        while not stop.is_set():
            t = time.perf_counter() - t_start
            try:
                s = smi_sample(amdsmi, h)
                s["t"] = t
                s["dt_read_ms"] = (time.perf_counter() - t_start - t) * 1e3
                samples.append(s)
            except Exception as exc:  # noqa: BLE001
                samples.append({"t": t, "error": repr(exc)})
                return
            stop.wait(args.smi_ms * 1e-3)

    th = None
    if amdsmi is not None:
        th = threading.Thread(target=sampler, daemon=True)
        th.start()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        w0 = time.perf_counter()
        fn()
        host = time.perf_counter() - w0
        e1.record(stream)
        return e0, e1, host

    reps = []
    i = 0
    graph_steps = 768  # a whole number of 32-buffer passes and 8-step ring periods
    while time.perf_counter() - t_start < args.seconds:
        rec = {"i": i, "t": time.perf_counter() - t_start}
        e0, e1, host = timed(lambda: eng.step_many(ring.data_ptr(), n, nbuf, args.steps))
        probing = args.probe_every and i % args.probe_every == args.probe_every - 1
        if probing:  # while the queued steps are still running: the shader clock under THIS load
            mhz, cpf = C.c_double(), C.c_double()
            if probe.clock_probe(20000, C.byref(mhz), C.byref(cpf)) == 0:
                rec["sclk_mhz_under_load"] = mhz.value
                rec["cycles_per_dependent_fma"] = cpf.value
        eng.sync()
        rec["eager_us"] = e0.elapsed_time(e1) * 1e3 / args.steps
        rec["host_enqueue_us"] = host * 1e6 / args.steps
        if probing:
            e0, e1, _ = timed(lambda: eng.step_many(ring.data_ptr(), n, nbuf, graph_steps, use_graph=True))
            eng.sync()
            rec["graph_us"] = e0.elapsed_time(e1) * 1e3 / graph_steps
            us = C.c_double()
            if lib.gymrs_tool_copy_probe(0, n * 17 // 16 * 16, n * 21 // 16 * 16, 200, 1, C.byref(us)) == 0:
                rec["copy_same_footprint_us"] = us.value
            if probe.alu_probe(4000, C.byref(us)) == 0:
                rec["alu_kernel_us"] = us.value
            # one more eager repetition right behind the probes (did THEY disturb anything?)
            e0, e1, _ = timed(lambda: eng.step_many(ring.data_ptr(), n, nbuf, args.steps))
            eng.sync()
            rec["eager_after_probes_us"] = e0.elapsed_time(e1) * 1e3 / args.steps
        reps.append(rec)
        i += 1
    stop.set()
    if th:
        th.join(timeout=1.0)
    eng.close()

    eager = [r["eager_us"] for r in reps]
    med = statistics.median(eager)
    fast = [r for r in reps if r["eager_us"] <= med * 1.03]
    slow = [r for r in reps if r["eager_us"] > med * 1.05]

    def col(rows, key):
        v = [r[key] for r in rows if key in r]
        return {"n": len(v), "median": statistics.median(v), "min": min(v), "max": max(v)} if v else None

    def smi_near(rows, key):
        if not samples or "error" in samples[-1] and len(samples) == 1:
            return None
        vals = []
        for r in rows:
            near = [s.get(key) for s in samples if abs(s["t"] - r["t"]) < 0.02 and isinstance(s.get(key), (int, float))]
            vals += near
        return {"n": len(vals), "median": statistics.median(vals), "min": min(vals), "max": max(vals)} if vals else None

    keys = ("eager_us", "host_enqueue_us", "graph_us", "copy_same_footprint_us", "alu_kernel_us", "sclk_mhz_under_load", "eager_after_probes_us")
    smi_keys = sorted({k for s in samples for k in s if k not in ("t", "dt_read_ms", "error")})
    # the longest run of consecutive slow repetitions
    longest, run = 0, 0
    for r in reps:
        run = run + 1 if r["eager_us"] > med * 1.05 else 0
        longest = max(longest, run)
    # drift: first 5 repetitions vs the rest (the driver's 5 repetitions came right after a short warm-up)
    out = {
        "tag": args.tag, "cpu_affinity": affinity, "seconds": args.seconds, "steps_per_repetition": args.steps, "smi": "on" if th else str(h),
        "repetitions": len(reps), "eager_median_us": med, "eager_min_us": min(eager), "eager_max_us": max(eager),
        "eager_p10_p90_us": [sorted(eager)[len(eager) // 10], sorted(eager)[len(eager) * 9 // 10]],
        "first_10_eager_us": [round(x, 3) for x in eager[:10]],
        "slow_repetitions": len(slow), "longest_slow_run": longest,
        "fast": {k: col(fast, k) for k in keys}, "slow": {k: col(slow, k) for k in keys},
        "smi_fast": {k: smi_near(fast, k) for k in smi_keys}, "smi_slow": {k: smi_near(slow, k) for k in smi_keys},
        "smi_read_ms_median": statistics.median([s["dt_read_ms"] for s in samples if "dt_read_ms" in s]) if samples and "dt_read_ms" in samples[0] else None,
        "n_samples": len(samples),
    }
    detail = os.environ.get("SLOW_MODE_DETAIL")
    if detail:
        Path(detail).write_text(json.dumps({"reps": reps, "samples": samples}))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
