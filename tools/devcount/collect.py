#!/usr/bin/env python
"""Collects profiles/devcount_traffic.json: fabric bytes per launch of FREE-RUNNING launches, per BASELINE config and call shape
(bench.PATHS), by device-wide counter sampling (chain_traffic.py under the rocprofiler-sdk tool libgymrs_devcount.so).

    python tools/devcount/collect.py [--only cartpole_2p20,...]        # on the GPU box; ~1 minute

Two processes per config (FETCH_SIZE and WRITE_SIZE each need the TCC counters to themselves); bytes by the calibration every
process takes itself from a 1 GiB copy (FETCH_SIZE reads exactly half of a wide streaming read on gfx950, WRITE_SIZE exactly:
the guide's correction, measured instead of assumed).  The file carries the hash of the kernel sources: bench.py prints a figure
only next to the kernels it was taken with."""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

CONFIGS = {"cartpole_2p20": ("cartpole", 1 << 20, 32, 20000)}
for name, (env, n, nbuf) in bench.EXTRA_CONFIGS.items():
    CONFIGS[name] = (env, n, nbuf, max(400, int(20000 * (1 << 20) / n)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=str(ROOT / "profiles" / "devcount_traffic.json"))
    args = ap.parse_args()
    tool = Path(__file__).resolve().parent / "libgymrs_devcount.so"
    sha = bench.kernel_source_sha16()
    out = {"kernel_source_sha16": sha,
           "how": "tools/devcount/collect.py: rocprofiler-sdk device counting service (agent-wide FETCH_SIZE / WRITE_SIZE), two samples around undisturbed "
                  "launches of each call shape; bytes by a 1 GiB copy's calibration in the same process",
           "configs": {}}
    only = [x for x in args.only.split(",") if x]
    for name, (env, n, nbuf, steps) in CONFIGS.items():
        if only and name not in only:
            continue
        rec = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [sys.executable, str(Path(__file__).resolve().parent / "chain_traffic.py"), "--counters", ctr, "--env", env, "--n", str(n),
                   "--nbuf", str(nbuf), "--steps", str(steps)]
            res = subprocess.run(cmd, env=dict(os.environ, ROCP_TOOL_LIBRARIES=str(tool)), capture_output=True, text=True, timeout=900)
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            if res.returncode != 0 or not lines:
                print(f"{name} {ctr}: failed (rc {res.returncode}): {res.stderr[-400:]}", file=sys.stderr)
                rec = None
                break
            rec[ctr] = json.loads(lines[0])
        if not rec:
            continue
        entry = {"lanes": n, "action_buffers": nbuf, "steps_per_window": steps}
        for path, phase in (("per_step_visible", "hip"), ("chain", "chain_again")):
            f = rec["FETCH_SIZE"]["phases"][phase]
            w = rec["WRITE_SIZE"]["phases"][phase]
            fetch = f["bytes_per_step_by_copy_calibration"][0]
            write = w["bytes_per_step_by_copy_calibration"][0]
            entry[path] = {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                           "event_us_per_step_while_counting": [f["event_us_per_unit"], w["event_us_per_unit"]], "launches_in_window": f["units"]}
        entry["calibration"] = {"FETCH_SIZE_per_GiB": rec["FETCH_SIZE"]["calibration"]["FETCH_SIZE"]["counter_per_GiB_copied"],
                                "WRITE_SIZE_per_GiB": rec["WRITE_SIZE"]["calibration"]["WRITE_SIZE"]["counter_per_GiB_copied"]}
        entry["dispatcher"] = rec["FETCH_SIZE"].get("aql")
        out["configs"][name] = entry
        print(name, {p: round(entry[p]["bytes_per_launch"] / 1e6, 2) for p in ("per_step_visible", "chain")}, "MB per launch", flush=True)
    Path(args.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
