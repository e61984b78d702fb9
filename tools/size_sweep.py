#!/usr/bin/env python
"""Developer tool: per-launch time of the step kernel and of the in-library copy probe over batch sizes (one GPU).

    python tools/size_sweep.py [--env 0] [--sizes 18,20,21,22,23,24] [--nt 0|1|2]

Prints, per size: step us, algorithmic GB/s, copy-probe us at the same footprint (plain and non-temporal), ratio.
"""
import argparse
import ctypes as C
import importlib
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

gymrs = importlib.import_module("gym-rs_amd")
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("gymrs_copy_probe_tool", ROOT / "tools" / "copy_probe" / "build.py")
probe = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(probe)  # the copy yardstick: a tool of its own since round 5 (tools/copy_probe)
BYTES = {0: (17, 21), 1: (9, 13), 2: (12, 25)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", type=int, default=0)
    ap.add_argument("--sizes", default="18,20,21,22,23,24")
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--vec", type=int, default=4)
    ap.add_argument("--nbuf", type=int, default=8)
    args = ap.parse_args()
    lib = gymrs.load_library()
    rd, wr = BYTES[args.env]
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | (gymrs.TIME_LIMIT if args.env == 2 else 0)
    esz = 4 if args.env == 2 else 1
    for lg in [int(x) for x in args.sizes.split(",")]:
        n = 1 << lg
        eng = gymrs.BatchedEngine(args.env, n, flags=flags)
        eng.set_tuning(args.vec, args.nt)
        ring = torch.empty(args.nbuf * n * esz, dtype=torch.uint8, device="cuda:0")
        for b in range(args.nbuf):
            eng.fill_actions(ring.data_ptr() + b * n * esz, seed=1, t=b)
        eng.reset(seed=0)
        st = torch.cuda.ExternalStream(eng.stream, device="cuda:0")
        steps = max(50, min(3000, int(2e10 / (n * 40))))
        eng.step_many(ring.data_ptr(), n * esz, args.nbuf, steps)
        eng.sync()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            eng.step_many(ring.data_ptr(), n * esz, args.nbuf, steps)
            e1.record(st)
            eng.sync()
            best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
        import json

        elided = 4 if (args.env == 0 and json.loads(eng.env_json(0))["gymrs"].get("reward_store_elided")) else 0  # CartPole from 128 MiB per step on
        eng.close()
        del ring
        torch.cuda.empty_cache()
        out = []
        chained = 2 if os.environ.get("GYMRS_AQL", "1") != "0" else 0  # the copy is submitted the way the steps are (chains unless GYMRS_AQL=0)
        for hint in (0, 1, 4):  # none | loads and stores | stores only
            us = probe.copy_probe(0, n * rd // 16 * 16, n * (wr - elided) // 16 * 16, max(20, steps // 4), hint | chained)
            out.append(us if us is not None else float("nan"))
        gbps = n * (rd + wr) / (best * 1e-6) / 1e9
        frac = "" if chained else f" ({gbps / 8000:.3f} of 8 TB/s)"  # (a chain's state is read out of the L2s at the small sizes: no HBM fraction)
        print(f"2^{lg:2d} lanes: step {best:9.2f} us  {gbps:7.1f} GB/s algorithmic{frac}   in-place copy, {'chain' if chained else 'HIP launches'}: "
              f"plain {out[0]:8.2f}  hinted {out[1]:8.2f}  stores hinted {out[2]:8.2f} us   step/copy {best / min(out):.3f}"
              + ("   (reward store elided: the copy moves 17 + 17 B per lane)" if elided else ""), flush=True)


if __name__ == "__main__":
    main()
