#!/usr/bin/env python
"""Round 6: gpurun_out/r06_* (written by tools/refresh_evidence.sh r06 on the GPU box) -> profiles/, with the three files that are assembled rather than copied
(the size sweep of six runs, the submission-by-size table's header) and BASELINE.md's round-6 rows.   python tools/publish_evidence_r06.py"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"


def clean(t):
    return "\n".join(l for l in t.splitlines() if "amdgpu.ids" not in l)


def main():
    subprocess.run(["bash", str(ROOT / "tools" / "publish_evidence.sh"), "r06", "r06"], cwd=ROOT, check=True)
    parts = []
    for env, name in ((0, "CartPole"), (1, "MountainCar"), (2, "Pendulum")):
        for aql, sub in ((1, "chains"), (0, "HIP launches (per-step visible)")):
            parts.append(f"== env {env} ({name}), {sub}\n" + clean((G / f"r06_size_sweep_env{env}_aql{aql}.log").read_text()))
            (P / f"r06_size_sweep_env{env}_aql{aql}.log").unlink(missing_ok=True)
    (P / "r06_size_sweep.log").write_text(
        "# profiles/r06_size_sweep.log -- step vs IN-PLACE copy of its own footprint over sizes, both call shapes, all three envs (tools/size_sweep.py, the round's evidence run\n"
        "# tools/refresh_evidence.sh r06, one box; the copy is tools/copy_probe's -- a tool of its own since round 5 -- submitted the way the steps are: launches of a chain /\n"
        "# HIP launches; no hint, loads + stores hinted, stores hinted; 8 action buffers; hashed non-zero words).  CartPole's kernel issues its loads ahead of its argument fetch since round 6 (profiles/r06_loads_before_arguments.log), the others are round 4's: compare profiles/r05_size_sweep.log.\n"
        + "\n".join(parts) + "\n")
    t = clean((G / "r06_submission_by_size.log").read_text())
    import re

    t = re.sub(r"\S*/gym-rs_amd/libgymrs_amd\.so ", "", t)
    (P / "r06_submission_by_size.log").write_text(
        "# profiles/r06_submission_by_size.log -- gymrs_step_many through its three submissions by env (0 CartPole, 1 MountainCar, 2 Pendulum) and size, the final tree, one box\n"
        "# (tools/refresh_evidence.sh r06 step 7; tools/step_timer.py --aql 0,1,2: one engine per submission timed alternately in one process; us per step by HIP events, median /\n"
        "# min / max of 5 repetitions; 8 action buffers).  GYMRS_AQL=0 HIP launches (per-step visible: acquire + release on every launch); 1 chains (acquire only, one release at\n"
        "# the end); 2 the engine's queue with HIP's header on every packet (per-step visible without the HIP runtime's host cost).\n" + t + "\n")
    (P / "r06_pytest_gpu_full.log").unlink(missing_ok=True)
    subprocess.run([sys.executable, str(ROOT / "tools" / "fill_baseline_r06.py"), str(P / "r06_bench_driver_form_full.json"), str(P / "r06_bench_in_process.json")], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
    print((G / "r06_sha.txt").read_text().strip())


if __name__ == "__main__":
    main()
