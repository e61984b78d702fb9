"""bench.py's output contract on a real GPU: one JSON line with the driver's fields, the roofline object and the
CPU baseline; the torchrun form with one rank goes through the RCCL all-reduce."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run(cmd, env=None):
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_common(d, n_gpus, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    lanes = d["config"]["total_lanes"]
    assert d["value"] == pytest.approx(lanes * steps / (d["ms_per_step"] * 1e-3 * steps), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["achieved"] == pytest.approx(r["bytes_per_launch"] / (r["launch_us"] * 1e-6) / 1e9)
    assert 0.05 < r["frac"] < 1.0
    # the event-derived launch time cannot exceed the wall time per step
    assert r["launch_us"] <= d["ms_per_step"] * 1e3 * 1.001


def test_default_form_prints_the_contract_line():
    d = run([sys.executable, "bench.py", "--steps", "300", "--warmup", "50", "--cpu-seconds", "1"])
    check_common(d, 1, 300, 50)
    assert d["config"]["lanes_per_gpu"] == 1 << 20 and "CartPole" in d["config"]["workload"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "env-steps/s" and c["value"] > 1e6 and c["sample"]
    assert d["value"] > 100 * c["value"]
    assert d["episodes"]["n_episodes"] > 0


def test_torchrun_form_with_one_rank_uses_rccl():
    env = dict(os.environ, GYMRS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", "29617", "bench.py", "--gpus", "1", "--steps", "200", "--warmup", "20", "--cpu-seconds", "0"], env=env)
    check_common(d, 1, 200, 20)


@pytest.mark.parametrize("env_name", ["mountain_car", "pendulum"])
def test_other_configs_run(env_name):
    d = run([sys.executable, "bench.py", "--env", env_name, "--steps", "100", "--warmup", "10", "--cpu-seconds", "0"])
    check_common(d, 1, 100, 10)
    assert d["episodes"]["n_episodes"] >= 0
