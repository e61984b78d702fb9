"""Builds tools/copy_probe/libgymrs_copy_probe.so -- the copy yardstick bench.py and tools/size_sweep.py compare a step launch with.
A measurement tool that links the library's dispatcher SOURCE against a code object of its own; nothing of it is in libgymrs_amd.so
or include/gymrs_amd.h (VERDICT r4 "next" #8).  hipcc cross-compiles gfx950 without a GPU; the .so travels to the GPU box with gpurun."""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
LIB = HERE / "libgymrs_copy_probe.so"


def _product_builder():
    spec = importlib.util.spec_from_file_location("gymrs_amd_build", ROOT / "gym-rs_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force: bool = False) -> Path:
    b = _product_builder()
    csrc = b.CSRC
    deps = [HERE / "copy_probe.hip", HERE / "copy_probe_aql.hip", HERE / "copy_probe_kernels.h", csrc / "gymrs_aql.hip", csrc / "gymrs_step_aql.hip",
            Path(__file__)] + b.HIP_HEADERS
    if not force and b._newer(LIB, deps):
        return LIB
    obj = HERE / "_obj"
    obj.mkdir(exist_ok=True)
    hipcc = b._hipcc()
    inc = [f"-I{ROOT / 'include'}", f"-I{csrc}", f"-I{HERE}"]
    flags = [f for f in b.HIPCC_FLAGS]
    # 1. the tool's code object (device only), under the name the dispatcher's .incbin asks for -- found in THIS object directory
    b._run([hipcc, "--cuda-device-only", "--no-gpu-bundle-output", *[f for f in flags if f != "-fPIC"], *inc, HERE / "copy_probe_aql.hip", "-o",
            obj / "gymrs_aql_kernels.hsaco"])
    # 2. the dispatcher's source against it (GYMRS_AQL_ENDS_ONLY: no step kernel is expected in the code object)
    b._run([hipcc, *flags, "-DGYMRS_AQL_ENDS_ONLY=1", *inc, f"-I{obj}", f"-Wa,-I{obj}", "-c", csrc / "gymrs_aql.hip", "-o", obj / "gymrs_aql_tool.o"])
    # 3. the probe itself, 4. link (-Bsymbolic: the tool's copy of the dispatcher is its own, whatever else the process has loaded)
    b._run([hipcc, *flags, *inc, "-c", HERE / "copy_probe.hip", "-o", obj / "copy_probe.o"])
    b._run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", obj / "copy_probe.o", obj / "gymrs_aql_tool.o", "-o", LIB,
            f"-L{b._rocm_lib(hipcc)}", "-lhsa-runtime64"])
    return LIB


_lib = None


def load():
    """ctypes handle with gymrs_tool_copy_probe(device, read_bytes, write_bytes, launches, mode, &us) -> int declared."""
    global _lib
    if _lib is None:
        import ctypes as C

        if not LIB.exists():
            raise OSError(f"{LIB} is missing: run `python tools/copy_probe/build.py` (or __graft_entry__.build())")
        lib = C.CDLL(str(LIB))
        lib.gymrs_tool_copy_probe.restype = C.c_int
        lib.gymrs_tool_copy_probe.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
        lib.gymrs_tool_copy_probe_error.restype = C.c_char_p
        lib.gymrs_tool_copy_probe_error.argtypes = []
        _lib = lib
    return _lib


def copy_probe(device: int, read_bytes: int, write_bytes: int, launches: int, mode: int):
    """Mean microseconds per launch, or None where that form is not available (the reason: load().gymrs_tool_copy_probe_error())."""
    import ctypes as C

    out = C.c_double()
    st = load().gymrs_tool_copy_probe(int(device), int(read_bytes), int(write_bytes), int(launches), int(mode), C.byref(out))
    return out.value if st == 0 else None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv[1:]))
