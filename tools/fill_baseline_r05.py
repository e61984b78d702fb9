#!/usr/bin/env python
"""Fills the R5_* placeholders of BASELINE.md's round-5 table from the full record of the driver-form bench run.

    python tools/fill_baseline_r05.py profiles/r05_bench_driver_form_full.json [profiles/r05_bench_in_process.json]
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def sci(x):
    m, e = f"{x:.3e}".split("e")
    return f"{m}e{int(e)}"


def f2(x):
    return "n/a" if x is None else f"{x:.2f}"


def main():
    d = json.loads(Path(sys.argv[1]).read_text())
    inproc = json.loads(Path(sys.argv[2]).read_text()) if len(sys.argv) > 2 else None
    rep = {}
    r = d["roofline"]
    rep.update({"R5_HEAD_VALUE": sci(d["value"]), "R5_HEAD_US": f"{r['launch_us']:.2f}", "R5_HEAD_FRAC": f2(r["frac"]), "R5_HEAD_MOVED": f2(r["frac_moved"]),
                "R5_HEAD_COUNTED": f2(r["frac_counted"]), "R5_HEAD_QUEUE": f2(r.get("queue_launch_us")),
                "R5_CPU16": sci(d["cpu_baseline"]["multi_thread"]["value"]), "R5_CPU": sci(d["cpu_baseline"]["value"])})
    c = d["paths"]["chain"]
    rep.update({"R5_CHAIN_VALUE": sci(c["value"]), "R5_CHAIN_US": f"{c['launch_us']:.2f}", "R5_CHAIN_FRAC": f2(c["roofline"]["frac"]),
                "R5_CHAIN_MOVED": f2(c["roofline"]["frac_moved"])})
    for key, name in (("MC", "mountain_car_2p20"), ("PD8", "pendulum_2p22_8_action_buffers"), ("PD", "pendulum_2p22"), ("C24", "cartpole_2p24_dram_resident"),
                      ("C25", "cartpole_2p25_hbm_streaming")):
        cf = d["configs"][name]
        rr = cf["roofline"]
        rep.update({f"R5_{key}_VALUE": sci(cf["value"]), f"R5_{key}_US": f"{cf['launch_us']:.2f}", f"R5_{key}_FRAC": f2(rr["frac"]), f"R5_{key}_MOVED": f2(rr["frac_moved"]),
                    f"R5_{key}_COUNTED": f2(rr["frac_counted"]), f"R5_{key}_QUEUE": f2(rr.get("queue_launch_us"))})
    rep["R5_INPROC_VALUE"] = sci(inproc["value"]) if inproc else "n/a"
    p = ROOT / "BASELINE.md"
    s = p.read_text()
    for k in sorted(rep, key=len, reverse=True):  # longest first: R5_PD8_* before R5_PD_*
        s = s.replace(k, rep[k])
    p.write_text(s)
    left = [w for w in s.split() if w.startswith("R5_")]
    print("filled", len(rep), "placeholders; left:", left[:5])


if __name__ == "__main__":
    main()
