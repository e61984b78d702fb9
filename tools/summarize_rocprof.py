#!/usr/bin/env python
"""Turns rocprofv3 rocpd databases (gpurun_out/...) into the small text/JSON summaries kept under profiles/.

    python tools/summarize_rocprof.py kernel  <results.db> <out.txt>     # --kernel-trace --stats summary
    python tools/summarize_rocprof.py traffic <fetch.db> <write.db> <env> <out.json>   # PMC FETCH_SIZE / WRITE_SIZE
    python tools/summarize_rocprof.py counters <results.db> <out.txt> [kernel-substring] [title]   # any --pmc pass: per-launch averages
    python tools/summarize_rocprof.py valu <results.db> <key> <lanes> <steps_per_launch> <sha16> <out.json>   # SQ_INSTS_VALU of the rollout kernel
"""
import json
import sqlite3
import statistics
import sys


def kernel_summary(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["rocprofv3 --kernel-trace --stats summary (durations in ns)", f"source: {db}", "",
             f"{'calls':>7} {'total_ns':>12} {'avg_ns':>10} {'min_ns':>8} {'max_ns':>8} {'pct':>6}  kernel"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"{n:>7} {tot:>12} {avg:>10.1f} {mn:>8} {mx:>8} {100.0 * tot / total:>6.2f}  {name}")
    ks = c.execute("select start, end from kernels where (name like '%step_kernel%' or name like 'gymrs_aql_cartpole%' or name like 'gymrs_aql_mountain%' or name like 'gymrs_aql_pendulum%') order by start").fetchall()
    if len(ks) > 20:
        skip = len(ks) // 10
        durs = [e - s for s, e in ks[skip:]]
        period = (ks[-1][0] - ks[skip][0]) / (len(ks) - 1 - skip)
        lines += ["", f"step_kernel: n={len(ks)} median duration {statistics.median(durs):.0f} ns, mean {statistics.mean(durs):.0f} ns, "
                      f"mean start-to-start period {period:.0f} ns (after skipping the first {skip})"]
        r = c.execute("select grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels where (name like '%step_kernel%' or name like 'gymrs_aql_cartpole%' or name like 'gymrs_aql_mountain%' or name like 'gymrs_aql_pendulum%') limit 1").fetchone()
        lines.append(f"step_kernel dispatch: grid_x={r[0]} workgroup_x={r[1]} lds={r[2]} vgpr={r[3]} sgpr={r[4]} scratch={r[5]}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def counter_avg(db, counter):
    c = sqlite3.connect(db)
    r = c.execute("select count(*), avg(value), min(value), max(value) from counters_collection where (kernel_name like "
                  "'%step_kernel%' or kernel_name like 'gymrs_aql_cartpole%' or name like 'gymrs_aql_mountain%' or name like 'gymrs_aql_pendulum%') and counter_name = ?", (counter,)).fetchone()
    return {"launches": r[0], "avg": r[1], "min": r[2], "max": r[3]}


def traffic(fetch_db, write_db, env, out):
    f = counter_avg(fetch_db, "FETCH_SIZE")
    w = counter_avg(write_db, "WRITE_SIZE")
    # /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are KB per dispatch; on gfx950
    # FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> double it.
    fetch_bytes = 2.0 * f["avg"] * 1024.0
    write_bytes = w["avg"] * 1024.0
    try:
        data = json.load(open(out))
    except Exception:
        data = {}
    data[env] = {
        "bytes_per_launch": fetch_bytes + write_bytes,
        "fetch_bytes": fetch_bytes, "write_bytes": write_bytes,
        "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests of wide coalesced reads as 64 B); WRITE_SIZE as reported",
        "source": [fetch_db, write_db],
    }
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(data[env], indent=1))


def counters(db, out, pattern="step_kernel", title=""):
    c = sqlite3.connect(db)
    rows = c.execute("select counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like ? "
                     "group by counter_name order by counter_name", (f"%{pattern}%",)).fetchall()
    k = c.execute("select kernel_name, grid_size, workgroup_size from counters_collection where kernel_name like ? limit 1", (f"%{pattern}%",)).fetchone()
    lines = [title or f"rocprofv3 --pmc per-launch averages, kernels matching '{pattern}'", f"source: {db}"]
    waves = None
    if k:
        try:
            waves = int(k[1]) // 64
            lines.append(f"kernel: {k[0][:160]}")
            lines.append(f"grid {k[1]} work-items, workgroup {k[2]} -> {waves} waves per launch")
        except Exception:
            waves = None
    lines.append("")
    for name, n, avg, mn, mx in rows:
        per_wave = f"  per_wave={avg / waves:10.1f}" if waves else ""
        lines.append(f"{name:28s} launches={n:5d} avg={avg:16.1f} min={mn:14.1f} max={mx:14.1f}{per_wave}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def valu(db, key, lanes, steps_per_launch, sha, out):
    c = sqlite3.connect(db)
    r = c.execute("select count(*), avg(value) from counters_collection where kernel_name like '%rollout_kernel%' and counter_name = 'SQ_INSTS_VALU'").fetchone()
    try:
        data = json.load(open(out))
        if data.get("kernel_source_sha16") != sha:
            data = {}
    except Exception:
        data = {}
    data["kernel_source_sha16"] = sha
    data[key] = {"SQ_INSTS_VALU_per_launch": r[1], "launches": r[0], "lanes": int(lanes), "steps_per_launch": int(steps_per_launch),
                 "source": db}
    json.dump(data, open(out, "w"), indent=1)
    print(key, data[key])


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_summary(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "counters":
        counters(*sys.argv[2:])
    elif sys.argv[1] == "valu":
        valu(*sys.argv[2:])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
