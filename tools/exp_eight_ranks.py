#!/usr/bin/env python
"""Developer experiment (profiles/r05_eight_ranks.log): `bench.py --gpus 8 --oversubscribe --n-envs 32768 --path <shape>` with the test's pinned schedule (8 processes sharing
ONE GPU, 16 queues); prints OK when the line's episode statistics are the CPU twin's for that schedule (warm-up 10, calibration 3 x 30, clear, 6 calls of 60 steps:
sum_length 94398043, n_episodes 5298280), the statistics otherwise.  argv[1] = per_step_visible (HIP launches only) | chain | both (the test's own command: 13 calls, want 204496222 / 11479920)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--oversubscribe", "--n-envs", "32768", "--cpu-seconds", "0", "--steps", "30", "--warmup", "10", "--no-probe",
       "--repetitions", "5", "--action-buffers", "8", "--path", sys.argv[1], "--full-out", f"/tmp/eight_{os.getpid()}.json"]
WANT = (94398043.0, 5298280.0) if sys.argv[1] != "both" else (204496222.0, 11479920.0)  # (both shapes: 13 calls of 60 steps after the clear)
env = dict(os.environ, GYMRS_BENCH_PASSES="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
res = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
try:
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    full = json.loads(Path(line["full"]).read_text())
    ep = full["episodes"]
    if (ep["sum_length"], ep["n_episodes"]) == WANT:
        print("OK")
    else:
        key = "chain" if sys.argv[1] == "both" else sys.argv[1]
        subs = sorted({str(r["paths"][key]["submission"])[:12] + "/" + str(r["paths"][key]["handover"])[:12] for r in full["ranks"]})
        per_rank = [(r["rank"], r.get("episodes")) for r in full["ranks"]]
        print(f"WRONG sum_length {ep['sum_length']:.0f} n_episodes {ep['n_episodes']:.0f} (want {WANT[0]:.0f} {WANT[1]:.0f}); submissions {subs}; per rank {per_rank}")
except Exception as exc:  # noqa: BLE001
    print("FAILED", res.returncode, repr(exc), res.stderr[-300:].replace("\n", " | "))
