#!/usr/bin/env python
"""Developer A/B timer (not part of the product or the tests): microseconds per step launch of ANY build of the
library, through the entry points every ABI version has.

    python tools/step_timer.py [--lib path/to/libgymrs_amd.so ...] [--env 0|1|2] [--n LANES] [--steps K] [--reps R]

Several --lib arguments are timed alternately in the same process on the same box (box-to-box spread is larger than
most kernel changes), R repetitions each; prints min / median per library.
"""
import argparse
import ctypes as C
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent


class Lib:
    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(str(path), mode=C.RTLD_LOCAL)
        L = self.lib
        L.gymrs_engine_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.gymrs_engine_destroy.argtypes = [C.c_void_p]
        L.gymrs_reset.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        L.gymrs_fill_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.gymrs_step_many.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        L.gymrs_sync.argtypes = [C.c_void_p]
        L.gymrs_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.gymrs_set_tuning.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.gymrs_last_error.restype = C.c_char_p

    def ck(self, st):
        if st != 0:
            raise RuntimeError(f"{self.path}: status {st}: {self.lib.gymrs_last_error().decode()}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--env", type=int, default=0)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--flags", type=int, default=-1)
    ap.add_argument("--vec", type=int, default=4)
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--nts", default="", help="comma-separated memory hints: one engine per (library, hint), timed alternately")
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--engine-first", type=int, default=0, help="create the engines before the action ring is allocated")
    ap.add_argument("--ring-2d", type=int, default=0)
    ap.add_argument("--wall", type=int, default=0, help="host wall clock (call + sync) instead of stream events")
    ap.add_argument("--all", type=int, default=0, help="print every repetition")
    ap.add_argument("--aql", default="", help="comma-separated GYMRS_AQL values (0 HIP launches, 1 chains, 2 the queue with HIP's header on every launch): "
                                              "one engine per (library, hint, value), timed alternately; default: the environment's")
    ap.add_argument("--hooks", default="", help="comma-separated gymrs_dev_set_hooks bit sets (e.g. 0,8: 8 = the chain's binary through HIP's queue): one engine each")
    ap.add_argument("--nbuf", type=int, default=32, help="action buffers in the ring (32 = bench.py's default)")
    args = ap.parse_args()
    libs = [Lib(p) for p in (args.lib or [ROOT / "gym-rs_amd" / "libgymrs_amd.so"])]
    flags = args.flags if args.flags >= 0 else (7 if args.env == 2 else 3)
    torch.cuda.init()
    nbuf = args.nbuf
    esz = 4 if args.env == 2 else 1
    ring = None
    if not args.engine_first:
        ring = torch.empty(nbuf * args.n * esz, dtype=torch.uint8, device="cuda:0")
    engines = []
    handles = []
    import os

    hook_sets = [int(x) for x in args.hooks.split(",") if x] or [None]
    aqls = [(q, hk) for q in ([x for x in args.aql.split(",") if x] or [None]) for hk in hook_sets]
    nts = [(int(x), q) for x in (args.nts.split(",") if args.nts else [str(args.nt)]) if x for q in aqls]
    libs = [lb for lb in libs for _ in nts]
    both = [m for _ in range(len(libs) // len(nts)) for m in nts]
    modes = [m[0] for m in both]
    aql_of = [m[1] for m in both]

    def with_aql(q):
        if q[0] is not None:
            os.environ["GYMRS_AQL"] = q[0]
    for lb in libs:
        h = C.c_void_p()
        lb.ck(lb.lib.gymrs_engine_create(args.env, args.n, 0, 0, None, flags, C.byref(h)))
        handles.append(h)
    if ring is None:
        ring = torch.empty(nbuf * args.n * esz, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
    print("ring at", hex(ring.data_ptr()), flush=True)
    for lb, h, nt in zip(libs, handles, modes):
        lb.ck(lb.lib.gymrs_set_tuning(h, args.vec, nt))
        lb.ck(lb.lib.gymrs_reset(h, 1, 0, None, None))
        for b in range(nbuf):
            lb.ck(lb.lib.gymrs_fill_actions(h, ring.data_ptr() + b * args.n * esz, 1, b))
        lb.ck(lb.lib.gymrs_sync(h))
        s = C.c_void_p()
        lb.ck(lb.lib.gymrs_get_stream(h, C.byref(s)))
        engines.append((lb, h, torch.cuda.ExternalStream(s.value, device="cuda:0")))
    keys = [f"{lb.path} nt={nt}" + (f" GYMRS_AQL={q[0]}" if q[0] is not None else "") + (f" hooks={q[1]}" if q[1] is not None else "") for lb, nt, q in zip(libs, modes, aql_of)]
    for (lb, h, _), q in zip(engines, aql_of):
        if q[1] is not None:
            lb.lib.gymrs_dev_set_hooks.argtypes = [C.c_void_p, C.c_uint32]
            lb.ck(lb.lib.gymrs_dev_set_hooks(h, q[1]))
    times = {k: [] for k in keys}
    host = {k: [] for k in keys}
    for (lb, h, st), q in zip(engines, aql_of):  # warm-up (clocks, caches)
        with_aql(q)
        lb.ck(lb.lib.gymrs_step_many(h, ring.data_ptr(), args.n * esz, nbuf, min(2000, args.steps), args.graph))
        lb.ck(lb.lib.gymrs_sync(h))
    for _ in range(args.reps):
        for key, (lb, h, st), q in zip(keys, engines, aql_of):
            with_aql(q)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            w0 = time.perf_counter()
            lb.ck(lb.lib.gymrs_step_many(h, ring.data_ptr(), args.n * esz, nbuf, args.steps, args.graph))
            host[key].append((time.perf_counter() - w0) * 1e6 / args.steps)
            e1.record(st)
            lb.ck(lb.lib.gymrs_sync(h))
            wall = (time.perf_counter() - w0) * 1e6 / args.steps
            times[key].append(wall if args.wall else e0.elapsed_time(e1) * 1e3 / args.steps)
    for p, ts in times.items():
        print(f"{statistics.median(ts):8.3f} us median  {min(ts):8.3f} min  {max(ts):8.3f} max   host enqueue {statistics.median(host[p]):6.3f} us/step   {p}", flush=True)
    if args.all:
        for p, ts in times.items():
            print("   reps:", " ".join(f"{t:.3f}" for t in ts), flush=True)
    for lb, h, _ in engines:
        lb.lib.gymrs_engine_destroy(h)


if __name__ == "__main__":
    sys.exit(main())
