#!/usr/bin/env python
"""Cold-start amplifier (round 6): the steady-state amplifier (tools/handover_amp.py) ran 355 000 chain calls of 8 processes on one GPU without ONE
wrong count, so round 5's wrong episode counter (3 occurrences in ~450 runs of `bench.py --gpus 8 --oversubscribe --n-envs 32768`) is not a
per-hand-over race: it belongs to a FRESH process.  This tool repeats exactly that: every round starts P fresh processes (no torch: numpy + ctypes, ~1 s
instead of 20 s per run) which all create their engine at the same moment and run the bench's schedule ONCE -- warm-up 10 steps, statistics, 3 x 30
steps, gymrs_stats_clear, 6 calls of 60 steps through HIP launches, 7 through chains, each call between HIP events on the engine's stream and a
barrier of all processes -- and compare the statistics with the CPU twin's.

On a mismatch the worker keeps the evidence (VERDICT r5 "next" #1b, the instrument): the raw per-wavefront episode slots, every lane's start tick and the
reset log of the wrong engine (gymrs_dev_peek: no fold, no snapshot), the same arrays of a SECOND engine of that process run through the same schedule
with a synchronise after every call, and the difference: which slots (hence which workgroups / XCDs), by how much, which lanes.

    python tools/coldstart_amp.py --seconds 300                 # parent
Uses oracle/ (the twin) as the CHECKER only: a developer / test tool."""
import argparse
import ctypes as C
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

PRE_CALLS = (10, 30, 30, 30)
CALL = 60
NBUF = 8


def twin_expectations(lanes, procs, hip_calls, chain_calls, seed=0):
    from oracle.bindings import Twin, TwinEngine

    gymrs = importlib.import_module("gym-rs_amd")
    tw = Twin()
    out = []
    for r in range(procs):
        e = TwinEngine(tw, 0, lanes, gymrs.engine.default_params(0), flags=3, gid0=r * lanes)
        e.reset(seed)
        ring = [e.fill_actions(1, b) for b in range(NBUF)]
        mid = None
        for i, call in enumerate(PRE_CALLS):
            for t in range(call):
                e.step(ring[t % NBUF])
            if i == 0:
                mid = e.stats().tolist()
        pre = e.stats().tolist()
        e.stats_clear()
        for _ in range(hip_calls + chain_calls):
            for t in range(CALL):
                e.step(ring[t % NBUF])
        out.append({"mid": mid, "pre": pre, "fin": e.stats().tolist()})
    return out


class Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hipEventSynchronize.argtypes = [C.c_void_p]
        self.lib.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def malloc(self, n):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), n) == 0
        return p.value

    def event(self):
        e = C.c_void_p()
        assert self.lib.hipEventCreate(C.byref(e)) == 0
        return e

    def sync_device(self):
        assert self.lib.hipDeviceSynchronize() == 0


def peek(lib, eng, what, dtype, count):
    buf = np.zeros(count, dtype=dtype)
    got = C.c_uint64()
    rc = lib.gymrs_dev_peek(eng._h, what, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.nbytes), C.byref(got))
    assert rc == 0, rc
    return buf[: got.value // buf.itemsize]


def worker(args):
    rank, lanes = args.rank, args.lanes
    shm = shared_memory.SharedMemory(name=args.shm)
    slots = np.ndarray((args.procs,), dtype=np.int64, buffer=shm.buf)
    state = {"i": 0}

    def meet():
        state["i"] += 1
        slots[rank] = state["i"]
        t0 = time.time()
        while int(slots.min()) < state["i"]:
            if time.time() - t0 > 120:
                raise TimeoutError("a peer never arrived")

    rec = {"rank": rank, "ok": False}
    try:
        want = json.loads(Path(args.expect).read_text())[rank]
        meet()  # every process is up (the launcher's rendezvous)
        t0 = time.time()
        gymrs = importlib.import_module("gym-rs_amd")
        hip = Hip()
        os.environ.pop("GYMRS_AQL", None)
        lib = gymrs.load_library()
        lib.gymrs_dev_peek.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        eng = gymrs.BatchedEngine(0, lanes, global_env_offset=rank * lanes, device=0, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS)
        rec["create_s"] = round(time.time() - t0, 3)
        ring = hip.malloc(NBUF * lanes)
        hip.sync_device()
        for b in range(NBUF):
            eng.fill_actions(ring + b * lanes, seed=1, t=b)
        stream = C.c_void_p(eng.stream)
        ev = [hip.event(), hip.event()]

        def run(e, how, k):
            os.environ["GYMRS_AQL"] = "0" if how == "hip" else "1"
            e.step_many(ring, lanes, NBUF, k)

        def schedule(e, timed):
            """The bench's flow on engine e; timed = with the events and barriers of the real run."""
            e.reset(seed=0)
            mid = None
            for i, call in enumerate(PRE_CALLS):
                run(e, "hip", call)
                e.sync()
                if i == 0:
                    mid = e.stats()
                if timed:
                    meet()
            e.stats_clear()
            for how, n_calls in (("hip", args.hip_calls), ("chain", args.chain_calls)):
                for j in range(n_calls):
                    if timed:
                        meet()
                        hip.sync_device()
                        hip.lib.hipEventRecord(ev[0], stream)
                    run(e, how, CALL)
                    if timed:
                        hip.lib.hipEventRecord(ev[1], stream)
                        hip.sync_device()
                        meet()
                    else:
                        e.sync()
            e.sync()
            return mid, e.stats()

        mid, fin = schedule(eng, True)
        rec["run_s"] = round(time.time() - t0, 3)
        extras = json.loads(eng.env_json(0)).get("gymrs", {})
        rec["handover"] = extras.get("aql_handover")
        rec["chains"] = extras.get("aql_chains")
        rec["ok"] = bool(np.array_equal(mid, want["mid"]) and np.array_equal(fin, want["fin"]))
        if not rec["ok"]:
            rec["got"] = {"mid": mid.tolist(), "fin": fin.tolist()}
            rec["want"] = want
            rec["excess_episodes"] = fin[2] - want["fin"][2]
            rec["excess_length"] = fin[1] - want["fin"][1]
            # ---- post-mortem: raw arrays of the wrong engine against a second engine of this process run through the same schedule, call by call ----
            info = peek(lib, eng, 3, np.uint32, 8)
            bs = peek(lib, eng, 0, np.uint64, 2 * int(info[0]))[0::2].astype(np.int64)
            ep = peek(lib, eng, 1, np.uint32, lanes).astype(np.int64)
            ref = gymrs.BatchedEngine(0, lanes, global_env_offset=rank * lanes, device=0, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS)
            mid2, fin2 = schedule(ref, False)
            rec["rerun_ok"] = bool(np.array_equal(fin2, want["fin"]))
            bs2 = peek(lib, ref, 0, np.uint64, 2 * int(info[0]))[0::2].astype(np.int64)
            ep2 = peek(lib, ref, 1, np.uint32, lanes).astype(np.int64)
            d = bs - bs2
            nz = np.nonzero(d)[0]
            threads = 512 if (1 << 20) <= lanes < (1 << 22) else 256  # (step_threads_of, gymrs_kernels.h: CartPole's 512-work-item window)
            waves_per_wg = threads // 64
            rec["post_mortem"] = {
                "slots": int(info[0]), "slots_differing": int(nz.size), "slot_diff_sum": int(d.sum()),
                "first_differing_slots": [[int(i), int(d[i]), int(bs[i]), int(bs2[i])] for i in nz[:24]],
                "diff_by_xcd_assuming_wg_mod_8": [int(d[[i for i in range(d.size) if (i // waves_per_wg) % 8 == x]].sum()) for x in range(8)],
                "ep_start_lanes_differing": int(np.count_nonzero(ep != ep2)), "ep_start_diff_sum": int((ep - ep2).sum()),
                "ep_start_diff_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique((ep - ep2)[ep != ep2], return_counts=True))},
                "first_lanes": [int(i) for i in np.nonzero(ep != ep2)[0][:16]],
            }
            ref.close()
        eng.close()
    except Exception as exc:  # noqa: BLE001
        rec["error"] = repr(exc)
    finally:
        slots[rank] = 1 << 62
        shm.close()
    print(json.dumps(rec), flush=True)
    return 0


def parent(args):
    expect = Path(f"/tmp/coldstart_expect_{os.getpid()}.json")
    t0 = time.time()
    expect.write_text(json.dumps(twin_expectations(args.lanes, args.procs, args.hip_calls, args.chain_calls)))
    twin_s = time.time() - t0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("GYMRS_AQL", "GYMRS_AQL_HANDOVER", "GYMRS_AQL_FENCES", "GYMRS_AQL_SYNC"):
        env.pop(k, None)
    if args.handover != "auto":
        env["GYMRS_AQL_HANDOVER"] = args.handover
    rounds = runs = wrong = errors = 0
    bad, handovers, round_s = [], {}, []
    deadline = time.time() + args.seconds
    while time.time() < deadline and (args.rounds == 0 or rounds < args.rounds) and len(bad) < 8:
        shm = shared_memory.SharedMemory(create=True, size=8 * args.procs)
        np.ndarray((args.procs,), dtype=np.int64, buffer=shm.buf)[:] = 0
        t1 = time.time()
        procs = [subprocess.Popen([sys.executable, __file__, "--worker", "--rank", str(r), "--shm", shm.name, "--expect", str(expect)] + args.forward,
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(args.procs)]
        for r, p in enumerate(procs):
            try:
                out, err = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                out, err = p.communicate()
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            rec = json.loads(lines[-1]) if lines else {"rank": r, "error": f"rc {p.returncode}: {err[-300:]}"}
            runs += 1
            handovers[str(rec.get("handover"))[:12]] = handovers.get(str(rec.get("handover"))[:12], 0) + 1
            if "error" in rec:
                errors += 1
                bad.append(dict(rec, round=rounds))
            elif not rec["ok"]:
                wrong += 1
                bad.append(dict(rec, round=rounds))
        round_s.append(time.time() - t1)
        rounds += 1
        shm.close()
        try:
            shm.unlink()
        except FileNotFoundError:
            pass
    expect.unlink()
    print(json.dumps({"tool": "coldstart_amp", "procs": args.procs, "lanes": args.lanes, "handover": args.handover, "rounds": rounds, "process_runs": runs,
                      "wrong": wrong, "errors": errors, "seconds_per_round": round(float(np.median(round_s)), 2) if round_s else None, "twin_s": round(twin_s, 1),
                      "handovers": handovers, "bad": bad}), flush=True)
    return 1 if (wrong or errors) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=32768)
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--rounds", type=int, default=0)
    ap.add_argument("--hip-calls", type=int, default=6)
    ap.add_argument("--chain-calls", type=int, default=7)
    ap.add_argument("--handover", choices=("kernel", "sync", "auto"), default="auto")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--shm", default="")
    ap.add_argument("--expect", default="")
    args, _ = ap.parse_known_args()
    if args.worker:
        return worker(args)
    args.forward = list(sys.argv[1:])
    return parent(args)


if __name__ == "__main__":
    sys.exit(main())
