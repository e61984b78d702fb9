"""Builds the native parts in-tree.

  gym-rs_amd/libgymrs_amd.so        HIP kernels + C ABI, hipcc --offload-arch=gfx950 (the PRODUCT)
  oracle/build/libgymrs_oracle.so   f64 C restatement of the reference (TEST infrastructure)
  oracle/build/libgymrs_f32twin.so  host build of the product's shared physics header (TEST infra)

Run as `python gym-rs_amd/build.py [--force]`, or through `__graft_entry__.build()`.
hipcc cross-compiles for gfx950 without a GPU.  The built .so files are git-ignored but travel to
the GPU box with gpurun.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
ORACLE = ROOT / "oracle"

HIP_SOURCES = [CSRC / "gymrs_step_cartpole.hip", CSRC / "gymrs_step_mountain_car.hip", CSRC / "gymrs_step_pendulum.hip",
               CSRC / "gymrs_rollout.hip", CSRC / "gymrs_aux.hip", CSRC / "gymrs_engine.hip", CSRC / "gymrs_engine_io.hip",
               CSRC / "gymrs_aql.hip", CSRC / "gymrs_sharded.hip"]
# The per-step kernels once more as a stand-alone gfx950 code object (device-only compile), embedded into the library by
# gymrs_aql.hip (.incbin) and loaded through HSA by the engine's own AQL dispatcher.
AQL_KERNELS = CSRC / "gymrs_step_aql.hip"
# every header of the kernel library: the same glob bench.kernel_source_sha16() hashes (a hand-kept
# list had missed gymrs_json.h and gymrs_pcg64.h, so editing them did not rebuild the library)
HIP_HEADERS = sorted(CSRC.glob("gymrs_*.h")) + [ROOT / "include" / "gymrs_amd.h"]
LIB = PKG / "libgymrs_amd.so"

# -ffp-contract=off: the shared physics header spells out every fma; nothing else may be fused,
# or the GPU and its CPU f32 twin stop being bit-identical.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    # Deliver the first 12 dwords of kernel arguments (step_kernel: 4 state pointers, the action pointer, n)
    # in SGPRs at wave launch instead of behind an s_load round trip: -0.2 us per 2^20-lane step (measured).
    # The compiler keeps a compatible prologue for firmware without the feature.
    "-mllvm",
    "-amdgpu-kernarg-preload-count=12",
]


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(d).stat().st_mtime <= t for d in deps)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _rocm_lib(hipcc: str) -> str:
    """The ROCm library directory (libhsa-runtime64): $ROCM_PATH/lib, else next to the hipcc that is used, else /opt/rocm/lib."""
    for root in (os.environ.get("ROCM_PATH"), str(Path(hipcc).resolve().parent.parent), "/opt/rocm"):
        if root and (Path(root) / "lib").is_dir():
            return str(Path(root) / "lib")
    return "/opt/rocm/lib"


def _run(cmd, cwd=None):
    print("+", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], cwd=cwd, check=True)


def build_hip(force: bool = False, extra_flags=(), out: Path | None = None) -> Path:
    """One object per translation unit (compiled concurrently: the kernel tables take minutes), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    lib = Path(out) if out else LIB
    objdir = lib.parent / "_obj" / lib.stem
    hdr_deps = HIP_HEADERS + [Path(__file__)]
    if not force and not extra_flags and _newer(lib, HIP_SOURCES + [AQL_KERNELS] + hdr_deps):
        return lib
    objdir.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    hsaco = objdir / "gymrs_aql_kernels.hsaco"
    jobs, after = [], []
    hsaco_stale = force or bool(extra_flags) or not _newer(hsaco, [AQL_KERNELS] + hdr_deps)
    if hsaco_stale:
        jobs.append([hipcc, "--cuda-device-only", "--no-gpu-bundle-output", *[f for f in HIPCC_FLAGS if f != "-fPIC"], *extra_flags,
                     f"-I{ROOT / 'include'}", f"-I{CSRC}", AQL_KERNELS, "-o", hsaco])
    for src in HIP_SOURCES:
        obj = objdir / (src.stem + ".o")
        embeds = src.stem == "gymrs_aql"  # .incbin of the code object: compiled after it
        if force or extra_flags or not _newer(obj, [src] + hdr_deps) or (embeds and hsaco_stale):
            # (gymrs_aql.hip embeds the code object with .incbin "gymrs_aql_kernels.hsaco": found through -I, so no path travels in a -D string)
            cmd = [hipcc, *HIPCC_FLAGS, *extra_flags, f"-I{ROOT / 'include'}", f"-I{CSRC}", *([f"-I{objdir}", f"-Wa,-I{objdir}"] if embeds else []), "-c", src, "-o", obj]
            (after if embeds else jobs).append(cmd)
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        list(pool.map(_run, jobs))
    for cmd in after:
        _run(cmd)
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *(objdir / (s.stem + ".o") for s in HIP_SOURCES), "-o", lib, "-ldl",
          f"-L{_rocm_lib(hipcc)}", "-lhsa-runtime64"])
    return lib


def build_oracle(force: bool = False) -> Path:
    out = ORACLE / "build"
    if force and out.exists():
        shutil.rmtree(out)
    _run(["make", "-C", str(ORACLE), "all"])
    return out


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    force = "--force" in argv
    build_hip(force)
    build_oracle(force)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
