#!/usr/bin/env python
"""Rewrites the measured rows of BASELINE.md's round-6 table (between the r6-table markers) from the full record of the driver-form bench run.

    python tools/fill_baseline_r05.py profiles/r06_bench_driver_form_full.json [profiles/r06_bench_in_process.json] [profiles/r06_bench_driver_form_first_call_full.json]
"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def sci(x):
    m, e = f"{x:.3e}".split("e")
    return f"{m}e{int(e)}"


def f2(x):
    return "n/a" if x is None else f"{x:.2f}"


def row(label, lanes, shape, value, us, r, extra=""):
    return (f"| {label} | {lanes} | {shape} | {value} | {us:.2f} | **{f2(r['frac'])}** | {f2(r.get('frac_moved'))} | {f2(r.get('frac_counted'))} | "
            f"{f2(r.get('queue_launch_us'))} | {extra} |")


def main():
    d = json.loads(Path(sys.argv[1]).read_text())
    inproc = json.loads(Path(sys.argv[2]).read_text()) if len(sys.argv) > 2 else None
    first = json.loads(Path(sys.argv[3]).read_text()) if len(sys.argv) > 3 else None
    r = d["roofline"]
    cb = d["cpu_baseline"]
    rows = ["| config | lanes | call shape | env-steps/s | us per launch | **`frac`** | `frac_moved` | `frac_counted` | through the engine's queue with HIP's header "
            "(`GYMRS_AQL=2`), us | CPU baseline (f64 C restatement, 1 of 256 host cores) |", "|---|---|---|---|---|---|---|---|---|---|"]
    head_val = f"**{sci(d['value'])}**" + (f" (the round's first call, another box: {sci(first['value'])})" if first else "")
    rows.append(row("2 · CartPole-v1, f32, 1 GPU (**headline**)", "2^20", "per-step visible", head_val, r["launch_us"], r,
                    f"{sci(cb['value'])} steps/s ({cb['multi_thread']['cores']} threads: {sci(cb['multi_thread']['value'])})"))
    c = d["paths"]["chain"]
    rows.append(f"| 2 · the same, reported separately | 2^20 | chain | {sci(c['value'])} | {c['launch_us']:.2f} | {f2(c['roofline']['frac'])} of the L2s' 34.5 TB/s | "
                f"{f2(c['roofline']['frac_moved'])} | -- | | |")
    names = (("3 · MountainCar-v0, 1 GPU", "2^20", "mountain_car_2p20"), ("4 · Pendulum-v1 (spec-derived), 32 action buffers", "2^22", "pendulum_2p22"),
             ("4' · the same with 8 action buffers", "2^22", "pendulum_2p22_8_action_buffers"),
             ("2' · CartPole, DRAM-resident (reward store elided: 34 B)", "2^24", "cartpole_2p24_dram_resident"),
             ("2'' · CartPole, nothing fits the Infinity Cache", "2^25", "cartpole_2p25_hbm_streaming"))
    for label, lanes, name in names:
        cf = d["configs"][name]
        val = sci(cf["value"])
        if first and name == "mountain_car_2p20":
            fc = first["configs"][name]
            val += (f" / {cf['launch_us']:.2f} us on this box; the round's first call, another box: {sci(fc['value'])} / {fc['launch_us']:.2f} us (these 4 us HIP launches go "
                    f"host-bound on a slow launching thread -- 4.6-5.7 us on two other boxes of this round, `profiles/r05_visible_through_queue.log` -- the queue's "
                    f"{f2(cf['roofline'].get('queue_launch_us'))} us does not)")
        rows.append(row(label, lanes, "per-step visible", val, cf["launch_us"], cf["roofline"]))
    rows.append("| 5 · CartPole, 2^23 lanes over 8 GPUs | 8 x 2^20 | both | driver-run: `bench.py --gpus 8` starts its own 8 ranks (`sharder: \"process-per-gpu\"`); "
                "`bench.py --in-process --gpus 8` runs the C ABI's native sharder (one engine + one host thread per device, one grouped RCCL all-reduce): on this round's "
                f"one-GPU boxes `--in-process --gpus 1` reads {sci(inproc['value']) if inproc else 'n/a'}, 4 blocks sharing the GPU are bit-identical to one engine "
                "(`tests/test_gpu_sharded_native.py`) | | | | | | |")
    p = ROOT / "BASELINE.md"
    s = p.read_text()
    s = re.sub(r"(<!-- r6-table-begin[^\n]*-->\n).*?(<!-- r6-table-end -->\n)", lambda m: m.group(1) + "\n".join(rows) + "\n" + m.group(2), s, flags=re.S)
    p.write_text(s)
    print("\n".join(rows))


if __name__ == "__main__":
    main()
