#!/usr/bin/env python
"""Hand-over amplifier (VERDICT r5 "next" #1a): P processes share ONE GPU, each loops the schedule in which round 5's wrong episode count
appeared -- `bench.py --gpus 8 --oversubscribe --n-envs 32768`: reset, 100 untimed steps, gymrs_stats_clear, calls of 60 steps through HIP
launches and through gymrs_step_many chains, statistics -- hundreds of times per second instead of once per 20 s, and compares every
iteration's statistics with the CPU f32 twin's for that (rank, seed).  A 1-in-200-runs event becomes several per minute, or the number
of clean hand-overs says how rare it is.

    python tools/handover_amp.py --procs 8 --seconds 60 --mode both            # parent: starts the workers, prints one JSON summary line
    modes: both  = HIP launches before and after the clear, then chains (the bench's default run: where every occurrence was seen)
           chain = every call through chains          hip = every call through HIP launches (GYMRS_AQL=0: no chain anywhere)
    --handover kernel|sync|auto   (GYMRS_AQL_HANDOVER: asynchronous / synchronous / calibrated per engine)
    --aql 1|2                     (what "chain" calls use: 1 = fence-free chains, 2 = the queue with HIP's own header)
    --fences AR                   (GYMRS_AQL_FENCES, e.g. 22 = system-scope acquire + release on every chain packet)
    --lockstep                    (the workers meet at a shared-memory barrier before every iteration, like the bench's ranks)
    --check-clear                 (gymrs_stats right behind gymrs_stats_clear must read zero: the cheap detector of ADVICE r5)

Uses oracle/ (the twin) as the CHECKER only: a developer / test tool, never part of the product path."""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time
from multiprocessing import shared_memory
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PRE_CALLS = (10, 30, 30, 30)  # warm-up + priming + two calibration passes of `bench.py --steps 30 --warmup 10` with GYMRS_BENCH_PASSES pinned
CALL = 60                     # a repetition: 2 passes of 30 steps
NBUF = 8


def schedule(mode, calls_per_shape):
    """[(how, steps)] after the clear; how = 'hip' | 'chain'."""
    if mode == "both":
        return [("hip", CALL)] * calls_per_shape + [("chain", CALL)] * calls_per_shape
    return [(mode, CALL)] * (2 * calls_per_shape)


def expected_stats(twin_mod, gymrs, rank, lanes, seed, n_post_calls, flags=3, audit=False):
    """The twin's statistics for one iteration of (rank, seed): [after the warm-up, before the clear, after the run (, state, step results)]."""
    tw, TwinEngine = twin_mod
    e = TwinEngine(tw, 0, lanes, gymrs.engine.default_params(0), flags=flags, gid0=rank * lanes)
    e.reset(seed)
    ring = [e.fill_actions(1, b) for b in range(NBUF)]
    mid = None
    for i, call in enumerate(PRE_CALLS):
        for t in range(call):
            e.step(ring[t % NBUF])
        if i == 0:
            mid = e.stats().copy()  # (the bench reads the statistics once after the warm-up: the all-reduce's first contact)
    pre = e.stats().copy()
    e.stats_clear()
    for _ in range(n_post_calls):
        for t in range(CALL):
            e.step(ring[t % NBUF])
    if audit:
        return mid, pre, e.stats().copy(), e.get_state().copy(), tuple(a.copy() for a in e.get_result())
    return mid, pre, e.stats().copy()


def worker(args):
    import torch  # device buffers for the action ring; the HIP runtime of this process

    from oracle.bindings import Twin, TwinEngine

    gymrs = importlib.import_module("gym-rs_amd")
    rank, lanes = args.rank, args.lanes
    post = schedule(args.mode, args.calls)
    t0 = time.time()
    want = {s: expected_stats((Twin(), TwinEngine), gymrs, rank, lanes, s, len(post), args.flags, args.audit) for s in range(args.seeds)}
    t_twin = time.time() - t0
    shm = shared_memory.SharedMemory(name=args.shm) if args.shm else None
    slots = np.ndarray((args.procs,), dtype=np.int64, buffer=shm.buf) if shm else None

    def meet(i):
        if slots is None:
            return
        slots[rank] = i
        while int(slots.min()) < i:
            pass

    torch.cuda.set_device(0)
    os.environ.pop("GYMRS_AQL", None)
    def make_engine():
        # chains are opt-in (round 6): an engine that will run them sets its dispatcher up at creation (queue, self-check, hand-over timing), before the loop
        os.environ["GYMRS_AQL"] = "0" if args.mode == "hip" else str(args.aql)
        return gymrs.BatchedEngine(0, lanes, global_env_offset=rank * lanes, device=0, flags=args.flags)

    eng = make_engine()
    ring = torch.empty((NBUF, lanes), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    for b in range(NBUF):
        eng.fill_actions(ring[b].data_ptr(), seed=1, t=b)
    eng.sync()
    ptr, stride = ring.data_ptr(), ring.stride(0)
    aql_value = {"hip": "0", "chain": str(args.aql)}
    pre_how = "chain" if args.mode == "chain" else "hip"

    def run(e, how, k):
        os.environ["GYMRS_AQL"] = aql_value[how]
        e.step_many(ptr, stride, NBUF, k)

    bad, iters, handovers = [], 0, 0
    try:
        return _loop(args, gymrs, eng, make_engine, want, post, pre_how, run, meet, bad, t_twin)
    finally:
        if slots is not None:
            slots[rank] = 1 << 62  # nobody waits for a worker that has left (or died)
        if shm:
            shm.close()


def _loop(args, gymrs, eng, make_engine, want, post, pre_how, run, meet, bad, t_twin):
    rank = args.rank
    iters = handovers = 0
    trips = []  # calls that FAILED LOUDLY (a chain's XCD check, a hand-over that gave up): not wrong results, but the chain's premise did not hold
    meet(1)  # every worker has its engine (queues, self-check, calibration) before anybody loops
    deadline = time.time() + args.seconds
    while time.time() < deadline and len(bad) < 20:
        seed = iters % args.seeds
        mid, pre, fin = want[seed][:3]
        if args.lockstep:
            meet(2 + iters)
        try:
            eng.reset(seed=seed)
            for i, call in enumerate(PRE_CALLS):
                run(eng, pre_how, call)
                if i == 0:
                    eng.sync()
                    got = eng.stats()
                    if not np.array_equal(got, mid):
                        bad.append({"iter": iters, "seed": seed, "where": "after warm-up", "got": got.tolist(), "want": mid.tolist()})
            eng.sync()
            eng.stats_clear()
            if args.check_clear:
                got = eng.stats()
                if got[1] != 0 or got[2] != 0:
                    bad.append({"iter": iters, "seed": seed, "where": "right after stats_clear", "got": got.tolist()})
            for j, (how, k) in enumerate(post):
                if args.audit and j == 2:  # stream -> chain behind a snapshot LOAD: a call that is thrown away, then the same call again from the restored engine
                    blob = eng.snapshot()
                    run(eng, how, k)
                    eng.restore(blob)
                run(eng, how, k)
                if args.audit and j == 1:  # chain -> host copy -> stream write -> chain: the state read out and written back as it is
                    eng.set_state(eng.get_state())
                if args.audit and j == 3:  # a step-result read-out right behind a call
                    eng.get_step_result()
                if not args.no_sync_between:
                    eng.sync()
            if args.audit:  # state bits and the last step's reward / done / truncated against the twin, before the statistics
                state, result = want[seed][3], want[seed][4]
                got_state, got_result = eng.get_state(), eng.get_step_result()
                if not np.array_equal(got_state.view(np.uint32), state.view(np.uint32)):
                    bad.append({"iter": iters, "seed": seed, "where": "state", "lanes_differing": int(np.count_nonzero((got_state.view(np.uint32) != state.view(np.uint32)).any(axis=0)))})
                for name, g, w in zip(("reward", "done", "truncated"), got_result, result):
                    if not np.array_equal(np.asarray(g).view(np.uint8), np.asarray(w).view(np.uint8)):
                        bad.append({"iter": iters, "seed": seed, "where": name})
            got = eng.stats()
            if not np.array_equal(got, fin):
                bad.append({"iter": iters, "seed": seed, "where": "end", "got": got.tolist(), "want": fin.tolist(), "pre_clear": pre.tolist(),
                            "excess_episodes": got[2] - fin[2], "excess_length": got[1] - fin[1]})
        except gymrs.GymrsError as exc:  # loud: the engine said its arrays may be stale.  A fresh engine goes on (the old one keeps to HIP launches from here)
            trips.append({"iter": iters, "status": exc.status, "message": str(exc)[:200]})
            try:
                eng.sync()
            except gymrs.GymrsError:
                pass
            eng.close()
            eng = make_engine()
            if len(trips) >= 50:
                break
        iters += 1
        handovers += sum(1 for how, _ in post if how == "chain") + (len(PRE_CALLS) if pre_how == "chain" else 0)
    extras = json.loads(eng.env_json(0)).get("gymrs", {})
    print(json.dumps({"rank": rank, "iterations": iters, "chain_calls": handovers, "bad": bad, "trips": trips, "twin_s": round(t_twin, 1),
                      "handover": extras.get("aql_handover"), "dispatcher": extras.get("aql"), "chains": extras.get("aql_chains")}), flush=True)
    eng.close()
    return 0


def parent(args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("GYMRS_AQL", "GYMRS_AQL_HANDOVER", "GYMRS_AQL_FENCES", "GYMRS_AQL_SYNC"):
        env.pop(k, None)
    if args.handover != "auto":
        env["GYMRS_AQL_HANDOVER"] = args.handover
    if args.fences:
        env["GYMRS_AQL_FENCES"] = args.fences
    shm = shared_memory.SharedMemory(create=True, size=8 * args.procs)
    np.ndarray((args.procs,), dtype=np.int64, buffer=shm.buf)[:] = 0
    t0 = time.time()
    procs = []
    for r in range(args.procs):
        cmd = [sys.executable, __file__, "--worker", "--rank", str(r), "--shm", shm.name] + args.forward
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    recs, errs = [], []
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=args.seconds + 600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            errs.append(f"rank {r}: killed at the limit")
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            errs.append(f"rank {r}: rc {p.returncode}: " + " | ".join(ln for ln in err.splitlines() if "resource_tracker" not in ln and "warnings.warn" not in ln)[-600:])
        else:
            recs.append(json.loads(lines[-1]))
    shm.close()
    try:
        shm.unlink()
    except FileNotFoundError:  # (a worker's resource tracker may have removed the name already)
        pass
    bad = [dict(b, rank=r["rank"]) for r in recs for b in r["bad"]]
    trips = [dict(t, rank=r["rank"]) for r in recs for t in r.get("trips", [])]
    summary = {"tool": "handover_amp", "mode": args.mode, "handover": args.handover, "aql": args.aql, "fences": args.fences or None, "lockstep": args.lockstep,
               "check_clear": args.check_clear, "audit": args.audit, "flags": args.flags, "procs": args.procs, "lanes": args.lanes, "seconds": args.seconds, "wall_s": round(time.time() - t0, 1),
               "iterations": sum(r["iterations"] for r in recs), "chain_calls": sum(r["chain_calls"] for r in recs),
               "wrong_iterations": len(bad), "bad": bad[:12], "loud_failures": len(trips), "trips": trips[:6], "handovers_seen": sorted({str(r["handover"]) for r in recs}), "dispatchers": sorted({str(r["dispatcher"]) for r in recs}), "errors": errs}
    print(json.dumps(summary), flush=True)
    return 1 if (bad or errs) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--lanes", type=int, default=32768)
    ap.add_argument("--mode", choices=("both", "chain", "hip"), default="both")
    ap.add_argument("--handover", choices=("kernel", "sync", "auto"), default="auto")
    ap.add_argument("--aql", type=int, choices=(1, 2), default=1)
    ap.add_argument("--fences", default="")
    ap.add_argument("--calls", type=int, default=3, help="calls of 60 steps per call shape after the clear")
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--lockstep", action="store_true")
    ap.add_argument("--check-clear", action="store_true")
    ap.add_argument("--audit", action="store_true", help="every hand-over class per iteration (VERDICT r5 next #2): state read-out + write-back, snapshot + restore, step-result "
                                                         "read-out between the calls; state bits and step results compared with the twin as well")
    ap.add_argument("--flags", type=int, default=3, help="engine flags: 3 = AUTO_RESET | TRACK_STATS (the bench), 7 = + TIME_LIMIT (refresh kernels and the truncated memset between launches)")
    ap.add_argument("--no-sync-between", action="store_true", help="no gymrs_sync between the calls after the clear")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--shm", default="")
    args, _ = ap.parse_known_args()
    if args.worker:
        return worker(args)
    # what the workers need to know: everything but --worker / --rank / --shm
    args.forward = list(sys.argv[1:])
    return parent(args)


if __name__ == "__main__":
    sys.exit(main())
