"""CPU tests of the drop-in boundary and the host logic: the C-ABI library loads and exports every
symbol include/gymrs_amd.h declares; spaces and constants match the reference; the product fails
LOUDLY without a GPU (no CPU fallback); lane sharding + the statistics all-reduce over 2 ranks (gloo)."""
import ctypes as C
import os
import re
import spawn_server
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "gymrs_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gymrs_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_header_symbol(gymrs):
    lib = gymrs.load_library()
    names = header_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/gymrs_amd.h but not exported"
    from importlib import import_module

    sigs = import_module("gym-rs_amd._lib").SIGNATURES
    assert sorted(sigs) == names, "ctypes binding table and header disagree"
    assert lib.gymrs_abi_version() == 3
    # the copy yardstick is a TOOL since ABI 3 (tools/copy_probe), not something a gym-rs maintainer binds (VERDICT r4 "next" #8)
    assert "gymrs_copy_probe" not in names and not hasattr(lib, "gymrs_copy_probe")
    assert {"gymrs_sharded_create", "gymrs_sharded_step", "gymrs_sharded_stats", "gymrs_allreduce_stats_multi"} <= set(names)


def test_default_params_match_reference_constants(gymrs, golden):
    cp = gymrs.engine.default_params(gymrs.CARTPOLE)
    for k, v in golden("cartpole")["constants"].items():  # cartpole.rs:94-103
        assert getattr(cp, k) == v
    mc = gymrs.engine.default_params(gymrs.MOUNTAIN_CAR)
    for k, v in golden("mountain_car")["constants"].items():  # mountain_car.rs:344-351
        assert getattr(mc, k) == v
    pd = gymrs.engine.default_params(gymrs.PENDULUM)
    for k, v in golden("pendulum")["constants"].items():
        assert getattr(pd, k) == v


def test_spaces_match_reference(gymrs):
    lib = gymrs.load_library()
    n = C.c_uint32()
    assert lib.gymrs_action_space(0, C.byref(n), None, None) == 0 and n.value == 2  # cartpole.rs:114
    assert lib.gymrs_action_space(1, C.byref(n), None, None) == 0 and n.value == 3  # mountain_car.rs:362
    lo, hi, dim = (C.c_double * 4)(), (C.c_double * 4)(), C.c_int()
    assert lib.gymrs_observation_space(0, None, lo, hi, C.byref(dim)) == 0 and dim.value == 4
    # cartpole.rs:105-113: +-(4.8, inf, 0.41887902047863906, inf)
    assert list(hi) == [4.8, float("inf"), 0.41887902047863906, float("inf")] and list(lo) == [-v for v in hi]
    assert lib.gymrs_observation_space(1, None, lo, hi, C.byref(dim)) == 0 and dim.value == 2
    assert list(lo)[:2] == [-1.2, -0.07] and list(hi)[:2] == [0.6, 0.07]  # mountain_car.rs:353-354


def test_no_cpu_fallback_without_gpu(gymrs):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gymrs.GymrsError) as exc:
        gymrs.BatchedEngine(gymrs.CARTPOLE, 16)
    assert "no CPU fallback" in str(exc.value) or "HIP" in str(exc.value)
    with pytest.raises(gymrs.GymrsError):
        gymrs.CartPoleEnv()
    # the in-process sharder (ABI 3) too: its worker threads start, the first block's engine cannot be created, everything is torn down again and
    # the caller's thread gets the WORKER's message with the shard and its device in front (the error string is per thread in the library)
    with pytest.raises(gymrs.GymrsError) as exc:
        gymrs.ShardedEngine(gymrs.CARTPOLE, 4096, [0, 1])
    assert "shard 0 (device 0)" in str(exc.value) and "no CPU fallback" in str(exc.value)
    lib = gymrs.load_library()
    out = (C.c_double * 4)()
    assert lib.gymrs_allreduce_stats_multi(None, 2, out, None) == 1 and lib.gymrs_sharded_stats(None, out) == 1 and lib.gymrs_sharded_sync(None) == 1


def test_bad_arguments_are_status_codes_not_crashes(gymrs):
    lib = gymrs.load_library()
    h = C.c_void_p()
    assert lib.gymrs_engine_create(7, 16, 0, 0, None, 0, C.byref(h)) == 1  # unknown kind -> GYMRS_EINVAL
    assert b"unknown env kind" in lib.gymrs_last_error()
    assert lib.gymrs_engine_create(0, 0, 0, 0, None, 0, C.byref(h)) == 1
    assert lib.gymrs_engine_create(0, 16, 0, 0, None, 2, C.byref(h)) == 1  # TRACK_STATS without AUTO_RESET
    assert lib.gymrs_step(None, None) == 1 and lib.gymrs_sync(None) == 1
    assert lib.gymrs_engine_destroy(None) == 0


def test_shard_range_covers_batch(gymrs):
    for n, w in ((1 << 23, 8), (1000, 3), (5, 8)):
        spans = [gymrs.shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (o1, c1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + c1 == o2
    assert gymrs.shard_range(1 << 23, 8, 3) == (3 << 20, 1 << 20)
    with pytest.raises(ValueError):
        gymrs.shard_range(10, 2, 2)


WORKER = r"""
import os, sys, importlib
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
gymrs = importlib.import_module("gym-rs_amd")
from oracle.bindings import Twin, TwinEngine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
N, STEPS = 6000, 40
flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
off, cnt = gymrs.shard_range(N, world, rank)
tw = Twin()
P = gymrs.engine.default_params(0)
shard = TwinEngine(tw, 0, cnt, P, flags=flags, gid0=off)   # the f32 twin stands in for the GPU shard on CPU
shard.reset(11)
for t in range(STEPS):
    shard.step(shard.fill_actions(1, t))
st = torch.tensor(shard.stats(), dtype=torch.float64)
dist.all_reduce(st)                                          # the ONLY collective of the path: 4 doubles
state = shard.get_state()
gathered = [None] * world
dist.all_gather_object(gathered, (off, state))
if rank == 0:
    full = TwinEngine(tw, 0, N, P, flags=flags, gid0=0)     # the same batch unsharded
    full.reset(11)
    for t in range(STEPS):
        full.step(full.fill_actions(1, t))
    assert np.array_equal(st.numpy(), full.stats()), (st.numpy(), full.stats())
    cat = np.concatenate([s for _, s in sorted(gathered, key=lambda x: x[0])], axis=1)
    assert np.array_equal(cat.view(np.uint32), full.get_state().view(np.uint32)), "sharding changed a lane"
    print("SHARD_OK", st.tolist())
dist.destroy_process_group()
"""


def test_two_rank_sharding_and_allreduce_gloo(tmp_path):
    """world_size 2 on CPU (gloo): shard-invariance (global env ids feed the Philox counter) and
    all-reduce(sum) of per-shard statistics == statistics of the unsharded batch."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script), str(ROOT)]
    res = spawn_server.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "SHARD_OK" in res.stdout


def test_cpp_trait_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    """The C++ host side (include/gymrs_env.hpp) builds against the C ABI with plain g++ (no HIP headers, no
    torch types in the signatures); without a GPU it must stop with the library's own error, not fall back."""
    exe = tmp_path / "test_env_mirror"
    lib_dir = ROOT / "gym-rs_amd"
    spawn_server.run(["g++", "-std=c++17", "-O1", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp" / "test_env_mirror.cpp"),
                    "-o", str(exe), f"-L{lib_dir}", "-lgymrs_amd", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run is covered by tests/test_gpu_envs.py::test_cpp_trait_mirror")
    res = spawn_server.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert res.returncode != 0 and "no CPU fallback" in (res.stdout + res.stderr)


def test_header_is_plain_c_and_links_from_gcc(tmp_path):
    """The drop-in boundary is a C ABI: include/gymrs_amd.h must compile as pedantic C99 (no C++, no HIP or torch types
    in the signatures) and a gcc-built caller must link against the library."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "gymrs_amd.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    gymrs_cartpole_params p;
    gymrs_mountain_car_params m;
    double state[4];
    int dim = -1;
    if (gymrs_default_params(GYMRS_CARTPOLE, &p) != GYMRS_OK || gymrs_default_params(GYMRS_MOUNTAIN_CAR, &m) != GYMRS_OK) return 1;
    if (gymrs_params_from_json(GYMRS_CARTPOLE, "{\"gravity\":3.5,\"kinematics_integrator\":\"Other\",\"state\":{\"x\":1,\"x_dot\":2,\"theta\":3,\"theta_dot\":4}}",
                               &p, state, &dim) != GYMRS_OK) return 2;
    printf("%d %.17g %.17g %d %d %.1f\n", gymrs_abi_version(), p.gravity, p.theta_threshold_radians, p.kinematics_integrator, dim, state[3]);
    if (gymrs_params_from_json(GYMRS_CARTPOLE, "[1,2]", &p, NULL, NULL) != GYMRS_EINVAL) return 3;
    return strstr(gymrs_last_error(), "not a JSON object") ? 0 : 4;
}
''')
    exe = tmp_path / "caller"
    lib_dir = ROOT / "gym-rs_amd"
    spawn_server.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
                    f"-L{lib_dir}", "-lgymrs_amd", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    res = spawn_server.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.split() == ["3", "3.5", "0.20943951023931953", "1", "4", "4.0"]


def test_params_from_json_keeps_missing_keys_and_rejects_wrong_types(gymrs):
    p, state = gymrs.params_from_json(gymrs.MOUNTAIN_CAR, '{"force": 0.002, "gymrs": {"max_episode_steps": 321}, "unknown": [1, {"a": null}]}')
    d = gymrs.engine.default_params(gymrs.MOUNTAIN_CAR)
    assert p.force == 0.002 and p.max_episode_steps == 321 and p.gravity == d.gravity and state is None
    p, state = gymrs.params_from_json(gymrs.PENDULUM, '{"g": 9.81, "state": {"theta": 0.5, "theta_dot": null}}')
    assert p.g == 9.81 and state[0] == 0.5 and state[1] != state[1]  # null = a non-finite float in serde_json's output
    for bad in ('{"gravity": "x"}', '{"gravity": 1,}', "", '{"kinematics_integrator": "RK4"}', '{"state": 3}'):
        with pytest.raises(gymrs.GymrsError):
            gymrs.params_from_json(gymrs.CARTPOLE, bad)


def _ryu_layout(v: float) -> str:
    """serde_json's f64 text (ryu pretty layout), modelled on Python's shortest round-trip digits."""
    import math
    if not math.isfinite(v):
        return "null"
    if v == 0:
        return "-0.0" if math.copysign(1, v) < 0 else "0.0"
    mant, exp = f"{v:.17e}".split("e")  # placeholder for the sign; digits come from repr
    sign = "-" if v < 0 else ""
    r = repr(abs(v))
    if "e" in r:
        m, e = r.split("e")
        e = int(e)
    else:
        m, e = r, 0
    ip, _, fp = m.partition(".")
    digits = (ip + fp).lstrip("0")
    kk = len(ip) + e if ip.strip("0") else e - (len(fp) - len(fp.lstrip("0")))
    digits = digits.rstrip("0") or "0"
    n = len(digits)
    if n <= kk <= 16:
        return sign + digits + "0" * (kk - n) + ".0"
    if 0 < kk <= 16:
        return sign + digits[:kk] + "." + digits[kk:]
    if -5 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    return sign + digits[0] + ("." + digits[1:] if n > 1 else "") + "e" + str(kk - 1)


def test_json_numbers_are_laid_out_like_serde_json(tmp_path):
    """gymrs_json.h printed 10.0 as "1e1"; serde_json (ryu) prints plain decimals with a trailing ".0" for
    decimal exponents in [-5, 16) and exponent form otherwise.  Text-level check of the header, against known serde_json outputs
    and a Python model of the layout on random doubles."""
    import random
    import struct
    known = {10.0: "10.0", 200.0: "200.0", 100000.0: "100000.0", 9.8: "9.8", 0.02: "0.02", 0.0025: "0.0025", 1e16: "1e16",
             1e15: "1000000000000000.0", 1.5e-7: "1.5e-7", 1e-5: "0.00001", 1e-6: "1e-6", -2.4: "-2.4",
             0.20943951023931953: "0.20943951023931953", 1e300: "1e300", 5e-324: "5e-324", 0.1: "0.1", 1.0: "1.0", 0.0: "0.0",
             123456789012345678.0: "1.2345678901234568e17", 0.5: "0.5", 0.07: "0.07", -1.2: "-1.2", 0.001: "0.001"}
    rnd = random.Random(5)
    vals = list(known) + [struct.unpack("<d", struct.pack("<Q", rnd.getrandbits(64)))[0] for _ in range(3000)]
    vals += [rnd.uniform(-100, 100) for _ in range(500)] + [float(rnd.randint(-10**17, 10**17)) for _ in range(500)]
    vals = [v for v in vals if v == v and abs(v) != float("inf")]
    src = tmp_path / "num.cpp"
    src.write_text('#include "gymrs_json.h"\n#include <cstdio>\n#include <cstdlib>\nint main(int, char** a){ FILE* f = fopen(a[1], "rb"); double v; '
                   'while (fread(&v, 8, 1, f) == 1) std::printf("%s\\n", gymrs::json::number(v).c_str()); return 0; }\n')
    exe = tmp_path / "num"
    spawn_server.run(["g++", "-std=c++17", "-O1", f"-I{ROOT / 'gym-rs_amd' / 'csrc'}", str(src), "-o", str(exe)], check=True)
    data = tmp_path / "vals.bin"
    data.write_bytes(b"".join(struct.pack("<d", v) for v in vals))
    out = spawn_server.run([str(exe), str(data)], capture_output=True, text=True, check=True).stdout.split()
    assert len(out) == len(vals)
    for v, text in zip(vals, out):
        assert float(text) == v, (v, text)           # round-trips
        assert text == _ryu_layout(v), (v, text, _ryu_layout(v))
        if v in known:
            assert text == known[v], (v, text)


def test_library_embeds_the_chain_code_object_and_bench_names_its_kernels():
    """The stand-alone gfx950 code object of the per-step kernels (gymrs_step_aql.hip) is embedded into the library (.incbin in
    gymrs_aql.hip) for the engine's own AQL dispatcher: an ELF for amdgcn with the C-named kernels in it.  bench.py's
    roofline.kernel names the variant the engine's chain_hint_bits picks for a size."""
    blob = (ROOT / "gym-rs_amd" / "libgymrs_amd.so").read_bytes()
    # every flag set of the 4-lane launch table x every hint variant (the names gymrs_engine.hip's aql_kernel_name asks for), + the chain's own three
    names = [b"gymrs_aql_wait_flag", b"gymrs_aql_set_flag", b"gymrs_aql_selfcheck"]
    for env, sizes in (("cartpole", (512, 256)), ("mountain_car", (256,)), ("pendulum", (256,))):
        for threads in sizes:
            for flags in (0, 1, 3, 4, 5, 7):
                for hint in ("nt", "o", "so", "pl"):
                    names.append(f"gymrs_aql_{env}_f{flags}_t{threads}_{hint}".encode())
    assert len(names) == 99
    for name in names:
        assert blob.count(name + b".kd") >= 1, name
    # the copy yardstick's kernels live in the tool's own code object (tools/copy_probe), not in the product library
    assert blob.count(b"gymrs_aql_copy_probe") == 0 and blob.count(b"copy_probe_kernel") == 0
    tool = ROOT / "tools" / "copy_probe" / "libgymrs_copy_probe.so"
    if tool.exists():
        tblob = tool.read_bytes()
        for name in [b"gymrs_aql_copy_probe_" + h + i for h in (b"pl", b"nt", b"st") for i in (b"", b"1")] + names[:3]:
            assert tblob.count(name + b".kd") >= 1, name
        assert tblob.count(b"gymrs_aql_cartpole") == 0
    # (which variant a launch used is reported by the engine itself: gymrs_env_json(...)["gymrs"]["last_launch"], bench.py's roofline.kernel)


def test_bench_rooflines_are_fractions_of_the_bound_that_applies():
    """bench.roofline_of (host logic), ONE meaning per key on every leg (VERDICT r4 "next" #2): `frac` = bytes moved BY CONSTRUCTION / time / peak,
    `frac_counted` = the contract's algorithmic bytes, `frac_moved` = fabric bytes of the counter file / time / 8 TB/s.  The per-step-visible shape is held
    to the HBM roofline; a chain to the L2s while the fabric sees less than 0.9 x the bytes by construction (or nothing is on file and the arrays are small)."""
    import bench

    n, b = 1 << 20, 38
    vis = bench.roofline_of("per_step_visible", n, b, 6.4, {"bytes_per_launch": 26.1e6, "fetch_bytes": 4.4e6, "write_bytes": 21.7e6}, None, "k", "sha")
    assert vis["bound"] == "hbm" and vis["peak"] == 8000.0 and vis["frac"] == pytest.approx(n * b / 6.4e-6 / 1e9 / 8000.0) and vis["frac"] < 1.0
    assert vis["frac_counted"] == vis["frac"] and vis["achieved"] == pytest.approx(vis["bytes_per_launch"] / 6.4e-6 / 1e9)
    assert vis["traffic"] == 26.1e6 and vis["frac_moved"] == pytest.approx(26.1e6 / 6.4e-6 / 1e9 / 8000.0) and vis["hbm_bound"] is False
    assert vis["traffic_from"].startswith("file:profiles/")  # a replay of a counter run, not a measurement of this run
    ch = bench.roofline_of("chain", n, b, 4.9, {"bytes_per_launch": 18.5e6}, None, "k", "sha")
    assert ch["bound"] == "l2" and ch["peak"] == 34500.0 and 0 < ch["frac"] < 1.0 and ch["hbm_bound"] is False
    assert ch["frac_moved"] == pytest.approx(18.5e6 / 4.9e-6 / 1e9 / 8000.0)  # always against HBM: the fabric is what it crosses
    big = bench.roofline_of("chain", 1 << 24, b, 99.0, {"bytes_per_launch": 660e6}, None, "k", "sha")
    assert big["bound"] == "hbm" and big["hbm_bound"] is True and big["frac"] < 1.0
    nofile = bench.roofline_of("chain", n, b, 4.9, None, "why", "k", "sha")
    assert nofile["bound"] == "l2" and nofile["traffic"] is None and nofile["frac_moved"] is None and nofile["hbm_bound"] is None and nofile["traffic_note"] == "why"
    assert bench.roofline_of("chain", 1 << 24, b, 99.0, None, "why", "k", "sha")["bound"] == "hbm"
    # the 2^24-lane CartPole leg with its reward store elided: 34 B by construction -> frac 0.81, counters -> 0.85, the contract's 38 -> 0.91
    # (the three figures VERDICT r4 weak #5 recomputed from profiles/r04_*)
    el = bench.roofline_of("per_step_visible", 1 << 24, 38, 87.96, {"bytes_per_launch": 595.2e6}, None, "k", "sha", bench.moved_bytes("cartpole", {"reward_store_elided": True}))
    assert el["bytes_per_env_step"] == 34.0 and el["bytes_per_env_step_counted"] == 38
    assert (round(el["frac"], 2), round(el["frac_moved"], 2), round(el["frac_counted"], 2)) == (0.81, 0.85, 0.91) and el["hbm_bound"] is True
    # Pendulum counts 37 B per env-step and moves ~32.3 by construction: 0.875 x the algorithmic bytes on the fabric is NOT cache residency
    pend = bench.roofline_of("chain", 1 << 22, 37, 23.0, {"bytes_per_launch": 135.6e6}, None, "k", "sha", bench.MOVED_BYTES["pendulum"])
    assert pend["bound"] == "hbm" and pend["hbm_bound"] is True and pend["frac"] < pend["frac_counted"]
    mc = bench.roofline_of("per_step_visible", n, 22, 4.1, None, "why", "k", "sha", bench.MOVED_BYTES["mountain_car"])
    assert mc["frac"] == pytest.approx(n * 18 / 4.1e-6 / 1e9 / 8000.0) and mc["frac_counted"] == pytest.approx(n * 22 / 4.1e-6 / 1e9 / 8000.0)


def test_bench_prints_a_line_the_driver_can_keep(tmp_path):
    """bench.compact_line: the LAST stdout line stays under 4 KB whatever the full record holds (round 4's 35 KB line came back from the driver as
    `parsed: null`); it carries the contract's fields, `roofline` and `cpu_baseline`, small per-config records and the path of the full record."""
    import json

    import bench

    roof = bench.roofline_of("per_step_visible", 1 << 20, 38, 6.4, {"bytes_per_launch": 26.1e6, "fetch_bytes": 4.4e6, "write_bytes": 21.7e6}, None,
                             "HIP launch: gymrs::step_kernel<CartPoleT, 4, flags 3 | hint nt, 512 work-items>", "sha")
    roof["how"] = "prose " * 200
    path_rec = {"value": 1.6e11, "launch_us": 6.4, "roofline": roof, "what": "x" * 500, "wall_ms_per_repetition": [1.0] * 9}
    full = {"metric": "env-steps/sec (whole node), CartPole-v1 @ 2^20 envs per MI355X", "value": 163840000000.12345, "unit": "env-steps/s", "n_gpus": 8, "steps": 20,
            "warmup": 5, "ms_per_step": 0.0064000001, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CartPole-v1 @ 2^20 envs per GPU, f32, auto-reset, random policy", "call_shape": "per_step_visible", "lanes_per_gpu": 1 << 20,
                       "total_lanes": 8 << 20, "call_shape_note": "y" * 400, "parallelism": "lane-sharded x8"},
            "timing": {"repetitions": 9, "steps_per_repetition": 16000, "event_us_per_step": {"min": 6.3, "median": 6.4, "max": 6.5, "spread": 0.03},
                       "wall_ms_per_repetition": [100.0] * 9},
            "roofline": roof, "paths": {"per_step_visible": path_rec, "chain": dict(path_rec, roofline=bench.roofline_of("chain", 1 << 20, 38, 4.9, None, "n", "k", "sha"))},
            "configs": {name: {"value": 1e11, "launch_us": 88.0, "roofline": roof, "paths": {"per_step_visible": path_rec, "chain": path_rec}} for name in bench.EXTRA_CONFIGS},
            "ranks": [{"rank": r, "launch_us": 6.4, "paths": {"a": "z" * 300}} for r in range(8)], "ranks_agree": True,
            "cpu_baseline": {"value": 5.65e7, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "s" * 400, "multi_thread": {"value": 8.6e8, "cores": 16, "sample": "t" * 300}}}
    assert len(json.dumps(full)) > 20000
    where = bench.write_full_record(full, str(tmp_path / "full.json"))
    assert json.loads(where.read_text()) == full
    text = bench.compact_line(full, where)
    assert "\n" not in text and len(text) <= 2600 < bench.LINE_LIMIT, len(text)
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "full"):
        assert key in line, key
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    for key in ("bound", "achieved", "peak", "unit", "frac", "frac_moved", "frac_counted", "traffic", "kernel", "bytes_per_launch", "launch_us", "hbm_bound", "traffic_from"):
        assert key in line["roofline"], key
    assert line["roofline"]["frac"] == pytest.approx(roof["frac"], rel=1e-5) and "how" not in line["roofline"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1 and len(line["cpu_baseline"]["sample"]) <= 160
    assert set(line["configs"]) == set(bench.EXTRA_CONFIGS) and set(line["paths"]) == {"chain"}
    assert set(line["configs"]["mountain_car_2p20"]) == {"value", "launch_us", "bound", "frac", "frac_moved", "frac_counted"}
    assert line["config"]["workload"] == full["config"]["workload"] and "model" not in line["config"] and line["full"] == str(where)
    # a record that would still be too long sheds its optional blocks instead of growing past the limit
    full["config"]["workload"] = "w" * 3000
    fat = bench.compact_line(full, where)
    assert len(fat) <= bench.LINE_LIMIT and "roofline" in json.loads(fat) and "cpu_baseline" in json.loads(fat)


def test_committed_traffic_files_cover_both_call_shapes():
    """profiles/devcount_traffic.json (free-running launches, device-wide counters) is what bench.py's roofline.traffic quotes;
    profiles/pmc_traffic.json the per-dispatch figures beside it.  When a file belongs to the current kernel sources it must hold both call
    shapes (a renamed kernel once made the chain's record silently disappear)."""
    import json
    from pathlib import Path

    import bench

    sha = bench.kernel_source_sha16()
    checked = 0
    dev = json.loads((Path(bench.ROOT) / "profiles" / "devcount_traffic.json").read_text()) if (Path(bench.ROOT) / "profiles" / "devcount_traffic.json").exists() else {}
    if dev.get("kernel_source_sha16") == sha:
        assert "cartpole_2p20" in dev["configs"] and set(bench.EXTRA_CONFIGS) <= set(dev["configs"])
        for name, rec in dev["configs"].items():
            for path in ("per_step_visible", "chain"):
                assert rec[path]["bytes_per_launch"] > 0 and rec[path]["fetch_bytes"] >= 0 and rec[path]["write_bytes"] > 0, (name, path)
        checked += 1
    pmc = json.loads((Path(bench.ROOT) / "profiles" / "pmc_traffic.json").read_text())
    if pmc.get("kernel_source_sha16") == sha:
        for env in ("cartpole", "mountain_car", "pendulum"):
            for path in ("per_step_visible", "chain"):
                assert pmc[env][path]["bytes_per_launch"] > 0 and pmc[env][path]["raw"]["FETCH_SIZE"]["launches"] > 0, (env, path)
        checked += 1
    if not checked:
        pytest.skip("the committed traffic files belong to other kernel sources (bench.py drops them as stale)")

def test_every_environment_variable_the_library_reads_is_documented():
    """VERDICT r3 weak #11 (knob growth): one table in INTEGRATION.md, and nothing read that is not in it."""
    import re

    read = set()
    for src in (ROOT / "gym-rs_amd" / "csrc").glob("gymrs_*"):
        read |= set(re.findall(r'getenv\("(GYMRS_[A-Z0-9_]+)"\)', src.read_text()))
    assert "GYMRS_AQL" in read
    doc = (ROOT / "INTEGRATION.md").read_text()
    assert not [v for v in sorted(read) if v not in doc]


def test_cpu_baseline_threads_follow_the_cgroup_quota(tmp_path):
    """the multi-thread leg of cpu_baseline uses the CPUs this process may really run on -- its affinity mask, capped by the container's
    cgroup CPU quota (v2 cpu.max, v1 cfs quota / period) -- and says which, instead of a constant."""
    import bench

    (tmp_path / "cpu.max").write_text("350000 100000\n")
    assert bench.cgroup_cpu_quota(str(tmp_path)) == 3.5
    n, how = bench.usable_cpus(str(tmp_path))
    assert n == min(3, len(os.sched_getaffinity(0))) and ("cgroup CPU quota 3.5" in how or "affinity mask" in how)
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cgroup_cpu_quota(str(tmp_path)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("200000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cgroup_cpu_quota(str(v1)) == 2.0
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.cgroup_cpu_quota(str(v1)) is None and bench.usable_cpus(str(v1))[0] == len(os.sched_getaffinity(0))
