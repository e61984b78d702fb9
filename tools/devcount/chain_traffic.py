#!/usr/bin/env python
"""Fabric traffic of a FREE-RUNNING chain (VERDICT r3 "next" #1c): device-wide counter sampling, nothing serialised.

    ROCP_TOOL_LIBRARIES=$PWD/tools/devcount/libgymrs_devcount.so python tools/devcount/chain_traffic.py \
        --counters FETCH_SIZE [--env cartpole] [--n 1048576] [--steps 20000]

One counter set per process (a derived metric may need all of a block's hardware counters).  Phases, each bracketed by two
samples of the agent-wide counters:
    idle          nothing runs: what the profiler's own sampling and the idle device add
    copy          torch device-to-device copies of 1 GiB: KNOWN bytes, calibrates the counter's unit on this box
    hip           K per-step launches through HIP (GYMRS_AQL=0): every launch ends with a release fence
    chain         K steps as ONE gymrs_step_many chain (the engine's own AQL queue), free-running
Prints the counter delta per phase, per step, and -- with the copy's calibration -- bytes per step.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--counters", default="FETCH_SIZE")
    ap.add_argument("--env", default="cartpole")
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--nbuf", type=int, default=32)
    args = ap.parse_args()
    import torch

    gymrs = importlib.import_module("gym-rs_amd")
    dc = C.CDLL(str(Path(__file__).resolve().parent / "libgymrs_devcount.so"))
    dc.gymrs_devcount_error.restype = C.c_char_p
    names = [c for c in args.counters.split(",") if c]
    out = (C.c_double * len(names))()

    def sample():
        if dc.gymrs_devcount_sample(out, len(names)) < 0:
            raise SystemExit("sample: " + dc.gymrs_devcount_error().decode())
        return [out[i] for i in range(len(names))]

    kind = {"cartpole": 0, "mountain_car": 1, "pendulum": 2}[args.env]
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | (gymrs.TIME_LIMIT if kind == 2 else 0)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    eng = gymrs.BatchedEngine(kind, args.n, device=0, flags=flags)
    esz = 4 if kind == 2 else 1
    ring = torch.empty(args.nbuf * args.n * esz, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for b in range(args.nbuf):
        eng.fill_actions(ring.data_ptr() + b * args.n * esz, seed=1, t=b)
    eng.reset(seed=0)
    eng.step_many(ring.data_ptr(), args.n * esz, args.nbuf, min(2000, args.steps))  # warm-up
    eng.sync()
    src = torch.ones(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    torch.cuda.synchronize()

    if dc.gymrs_devcount_start(args.counters.encode()) != 0:
        raise SystemExit("start: " + dc.gymrs_devcount_error().decode())
    stream = torch.cuda.ExternalStream(eng.stream, device=dev)
    rec = {"counters": names, "env": args.env, "lanes": args.n, "steps": args.steps, "phases": {}}

    def phase(label, fn, units):
        torch.cuda.synchronize()
        a = sample()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        fn()
        e1.record(stream)
        eng.sync()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        b = sample()
        rec["phases"][label] = {"delta": [y - x for x, y in zip(a, b)], "raw_before": a, "raw_after": b, "units": units, "wall_ms": wall * 1e3,
                                "event_us_per_unit": e0.elapsed_time(e1) * 1e3 / max(units, 1)}

    phase("idle", lambda: time.sleep(0.05), 1)
    n_copies = 20
    phase("copy", lambda: [dst.copy_(src) for _ in range(n_copies)], n_copies)
    os.environ["GYMRS_AQL"] = "0"
    phase("hip", lambda: eng.step_many(ring.data_ptr(), args.n * esz, args.nbuf, args.steps), args.steps)
    os.environ["GYMRS_AQL"] = "1"
    phase("chain", lambda: eng.step_many(ring.data_ptr(), args.n * esz, args.nbuf, args.steps), args.steps)
    phase("chain_again", lambda: eng.step_many(ring.data_ptr(), args.n * esz, args.nbuf, args.steps), args.steps)
    phase("idle_after", lambda: time.sleep(0.05), 1)
    dc.gymrs_devcount_stop()
    ex = json.loads(eng.env_json(0))["gymrs"]
    rec["aql"] = {k: ex.get(k) for k in ("aql", "aql_handover", "aql_chains", "aql_launches")}
    # calibration: the copy moved 1 GiB each way per unit
    gib = float(1 << 30)
    for k, name in enumerate(names):
        per_copy = rec["phases"]["copy"]["delta"][k] / n_copies
        rec.setdefault("calibration", {})[name] = {"counter_per_GiB_copied": per_copy, "note": "1 GiB read + 1 GiB written per copy"}
        for ph in ("hip", "chain", "chain_again"):
            p = rec["phases"][ph]
            p.setdefault("counter_per_step", []).append(p["delta"][k] / p["units"])
            p.setdefault("bytes_per_step_by_copy_calibration", []).append(p["delta"][k] / p["units"] / per_copy * gib if per_copy else None)
    print(json.dumps(rec))
    eng.close()


if __name__ == "__main__":
    main()
