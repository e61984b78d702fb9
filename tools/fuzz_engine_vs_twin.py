#!/usr/bin/env python3
"""tools/fuzz_engine_vs_twin.py -- developer tool: random walks through the engine's entry points, the CPU f32 twin replaying every one.

Every case draws an env kind, a flag set, a lane count (log-uniform, ragged), an action ring and a sequence of operations --
gymrs_step_many calls of random length (short ones are HIP launches, long ones chains through the engine's own AQL dispatcher,
some as replays of captured HIP graphs), single gymrs_step / gymrs_step_host launches, fused rollouts, statistics reads and
clears, seeded resets, set_state, set_params with a new episode cap, clones that take over -- and after EVERY operation
compares state bits, the step result arrays and the statistics with the twin (Pendulum's returns, float sums taken per wavefront, to 1e-5).  The point is the interplay of the host-side state
machines (reset-log folds, the time-limit elision and its refreshes, chains that are closed and reopened in mid-call).

  python tools/fuzz_engine_vs_twin.py --cases 200 --seed 1        # ~1 minute on an MI355X
"""
import argparse
import importlib
import json
import os
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ops", type=int, default=14)
    ap.add_argument("--max-lanes", type=int, default=200_000)
    ap.add_argument("--min-lanes", type=int, default=65, help="(> 64: smaller engines keep their arrays in mapped host memory)")
    ap.add_argument("--verbose", type=int, default=0)
    args = ap.parse_args()
    import torch

    gymrs = importlib.import_module("gym-rs_amd")
    from oracle.bindings import Twin, TwinEngine

    twin = Twin()
    rng = random.Random(args.seed)
    totals = {"cases": 0, "ops": 0, "chains": 0, "aql_launches": 0, "elided": 0, "refreshes": 0}
    for case in range(args.cases):
        kind = rng.choice([0, 0, 1, 2])
        flags = rng.choice([0, 1, 3, 3, 4, 5, 7, 7])
        n = int(round(args.min_lanes * (args.max_lanes / args.min_lanes) ** rng.random()))
        nbuf = rng.randint(1, 9)
        gid0 = rng.choice([0, 12345, 1 << 33])
        p = gymrs.engine.default_params(kind)
        p.max_episode_steps = rng.choice([5, 17, 40, 200, 500])
        desc = f"case {case}: kind {kind} flags {flags} n {n} nbuf {nbuf} gid0 {gid0} cap {p.max_episode_steps}"
        log = [desc]
        eng = gymrs.BatchedEngine(kind, n, flags=flags, params=p, global_env_offset=gid0)
        tw = TwinEngine(twin, kind, n, p, flags=flags, gid0=gid0)
        seed0 = rng.randint(0, 1 << 30)
        eng.reset(seed=seed0)
        tw.reset(seed0)
        ring = torch.empty((nbuf, n), dtype=torch.float32 if kind == 2 else torch.uint8, device="cuda:0")
        aseed = rng.randint(1, 99)
        for b in range(nbuf):
            eng.fill_actions(ring[b].data_ptr(), seed=aseed, t=b)
        bufs = [tw.fill_actions(aseed, b) for b in range(nbuf)]
        stride = ring.stride(0) * ring.element_size()
        t_roll = 1000

        def check(what):
            ok = same(eng.get_state(), tw.get_state())
            res, ref = eng.get_step_result(), tw.get_result()
            ok = ok and same(res[0], ref[0]) and same(res[1], ref[1]) and (not (flags & 4) or same(res[2], ref[2]))
            if (flags & 3) == 3:
                a, b = eng.stats(), tw.stats()
                ok = ok and (np.allclose(a, b, rtol=1e-5) if kind == 2 else np.array_equal(a, b))
            if not ok:
                print("MISMATCH after", what)
                print("\n".join(log))
                print(json.loads(eng.env_json(0))["gymrs"])
                sys.exit(1)

        for _ in range(args.ops):
            op = rng.choices(["many_long", "many_short", "many_graph", "step", "step_host", "rollout", "stats_clear", "reset", "set_state", "set_params",
                              "clone", "sync", "hint"], weights=[8, 3, 2, 3, 1, 2, 1, 1, 1, 2, 1, 1, 2])[0]
            if op == "hint":  # (round 4) another memory hint = another instantiation of the same kernel, for HIP launches and for chains: never another result
                h = rng.randrange(4)
                log.append(f"set_tuning memory_hint {h}")
                eng.set_tuning(4, h)
                continue
            if op in ("many_long", "many_short", "many_graph"):
                graph = op == "many_graph"
                if graph and kind == 2 and (flags & 4):
                    continue  # (Pendulum's time limit is a host-computed kernel argument: gymrs_step_many refuses use_graph there)
                k = rng.randint(8, 150) if op != "many_short" else rng.randint(1, 7)
                log.append(f"step_many {k}" + (" use_graph" if graph else ""))
                eng.step_many(ring.data_ptr(), stride, nbuf, k, use_graph=graph)
                for t in range(k):
                    tw.step(bufs[t % nbuf])
            elif op == "step_host":
                b = rng.randrange(nbuf)
                log.append(f"step_host buf {b}")
                eng.step_host(bufs[b])
                tw.step(bufs[b])
            elif op == "step":
                b = rng.randrange(nbuf)
                log.append(f"step buf {b}")
                eng.step(ring[b].data_ptr())
                tw.step(bufs[b])
            elif op == "rollout":
                if not (flags & 1):
                    continue  # (the fused rollout is the caller loop WITH its reset-on-done)
                k = rng.randint(1, 40)
                log.append(f"rollout {k} t0 {t_roll}")
                eng.rollout(k, action_seed=aseed, action_t0=t_roll)
                for t in range(t_roll, t_roll + k):
                    tw.step(tw.fill_actions(aseed, t))
                t_roll += k
                if (flags & 3) == 3:  # (the rollout leaves the arrays as its last step would: state and statistics are comparable)
                    a, b = eng.stats(), tw.stats()
                    ok = same(eng.get_state(), tw.get_state()) and (np.allclose(a, b, rtol=1e-5) if kind == 2 else np.array_equal(a, b))
                else:
                    ok = same(eng.get_state(), tw.get_state())
                if not ok:
                    print("MISMATCH after rollout")
                    print("\n".join(log))
                    sys.exit(1)
                totals["ops"] += 1
                continue
            elif op == "stats_clear":
                if (flags & 3) != 3:
                    continue
                log.append("stats_clear")
                eng.stats_clear()
                tw.stats_clear()
            elif op == "reset":
                s = rng.randint(0, 1 << 30)
                log.append(f"reset {s}")
                eng.reset(seed=s)
                tw.reset(s)
                continue  # (no step result to compare right after a reset)
            elif op == "set_state":
                st = eng.get_state()
                m = rng.randint(1, min(n, 500))
                st[:, :m] = 0.01
                log.append(f"set_state first {m}")
                eng.set_state(st)
                tw.set_state(st)
                continue
            elif op == "set_params":
                p.max_episode_steps = rng.choice([3, 9, 25, 60, 200, 500])
                log.append(f"set_params cap {p.max_episode_steps}")
                eng.set_params(p)
                tw.set_params(p)
                continue
            elif op == "clone":
                log.append("clone takes over")
                other = eng.clone()
                x = json.loads(eng.env_json(0))["gymrs"]
                for key, name in (("chains", "aql_chains"), ("aql_launches", "aql_launches"), ("elided", "time_limit_elided_launches"),
                                  ("refreshes", "time_limit_refreshes")):
                    totals[key] += int(x.get(name, 0))
                eng.close()
                eng = other
            elif op == "sync":
                eng.sync()
                continue
            check(log[-1])
            totals["ops"] += 1
        x = json.loads(eng.env_json(0))["gymrs"]
        for key, name in (("chains", "aql_chains"), ("aql_launches", "aql_launches"), ("elided", "time_limit_elided_launches"),
                          ("refreshes", "time_limit_refreshes")):
            totals[key] += int(x.get(name, 0))
        eng.close()
        totals["cases"] += 1
        if args.verbose:
            print(desc, "ok", flush=True)
    print("fuzz ok:", totals)


if __name__ == "__main__":
    main()
