#!/usr/bin/env python
"""Turns rocprofv3 rocpd databases (gpurun_out/...) into the small text/JSON summaries kept under profiles/.

    python tools/summarize_rocprof.py kernel  <results.db> <out.txt>     # --kernel-trace --stats summary
    python tools/summarize_rocprof.py traffic <fetch.db> <write.db> <env> <out.json>   # PMC FETCH_SIZE / WRITE_SIZE
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
    ks = c.execute("select start, end from kernels where name like '%step_kernel%' order by start").fetchall()
    if len(ks) > 20:
        skip = len(ks) // 10
        durs = [e - s for s, e in ks[skip:]]
        period = (ks[-1][0] - ks[skip][0]) / (len(ks) - 1 - skip)
        lines += ["", f"step_kernel: n={len(ks)} median duration {statistics.median(durs):.0f} ns, mean {statistics.mean(durs):.0f} ns, "
                      f"mean start-to-start period {period:.0f} ns (after skipping the first {skip})"]
        r = c.execute("select grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels where name like '%step_kernel%' limit 1").fetchone()
        lines.append(f"step_kernel dispatch: grid_x={r[0]} workgroup_x={r[1]} lds={r[2]} vgpr={r[3]} sgpr={r[4]} scratch={r[5]}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def counter_avg(db, counter):
    c = sqlite3.connect(db)
    r = c.execute("select count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like "
                  "'%step_kernel%' and counter_name = ?", (counter,)).fetchone()
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


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_summary(sys.argv[2], sys.argv[3])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
