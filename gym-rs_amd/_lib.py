"""ctypes loader for ``libgymrs_amd.so`` (the C ABI declared in ``include/gymrs_amd.h``).

Fails loudly when the library is missing: the product path has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
ABI_VERSION = 3  # GYMRS_ABI_VERSION of include/gymrs_amd.h
_LIB = None

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); mirrors include/gymrs_amd.h one to one
SIGNATURES = {
    "gymrs_default_params": (C.c_int, [C.c_int, C.c_void_p]),
    "gymrs_action_space": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), f64p, f64p]),
    "gymrs_observation_space": (C.c_int, [C.c_int, C.c_void_p, f64p, f64p, C.POINTER(C.c_int)]),
    "gymrs_discrete_contains": (C.c_int, [C.c_uint64, C.c_uint64]),
    "gymrs_engine_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "gymrs_engine_destroy": (C.c_int, [C.c_void_p]),
    "gymrs_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gymrs_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, f32p, u64p]),
    "gymrs_reset_pcg64": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_double), u64p]),
    "gymrs_step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gymrs_step_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gymrs_step_many": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
    "gymrs_sync": (C.c_int, [C.c_void_p]),
    "gymrs_obs_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "gymrs_state_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "gymrs_reward_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_done_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_truncated_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_get_obs": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "gymrs_get_state": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "gymrs_set_state": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "gymrs_get_step_result": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gymrs_stats": (C.c_int, [C.c_void_p, f64p]),
    "gymrs_stats_clear": (C.c_int, [C.c_void_p]),
    "gymrs_stats_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gymrs_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gymrs_allreduce_stats": (C.c_int, [C.c_void_p, f64p]),
    "gymrs_engine_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_snapshot_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "gymrs_snapshot_save": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gymrs_snapshot_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gymrs_rollout_record": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]),
    "gymrs_rollout": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "gymrs_fill_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "gymrs_get_tick": (C.c_int, [C.c_void_p, u64p, u64p]),
    "gymrs_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gymrs_set_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gymrs_get_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gymrs_env_json": (C.c_int, [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, u64p]),
    "gymrs_params_from_json": (C.c_int, [C.c_int, C.c_char_p, C.c_void_p, f64p, C.POINTER(C.c_int)]),
    # one batch over several GPUs in one process
    "gymrs_allreduce_stats_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, f64p, C.POINTER(C.c_int)]),
    "gymrs_sharded_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "gymrs_sharded_destroy": (C.c_int, [C.c_void_p]),
    "gymrs_sharded_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "gymrs_sharded_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), u64p, u64p, C.POINTER(C.c_int)]),
    "gymrs_sharded_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, f32p, u64p]),
    "gymrs_sharded_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gymrs_sharded_step_many": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
    "gymrs_sharded_fill_actions": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint64, C.c_uint64]),
    "gymrs_sharded_rollout": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "gymrs_sharded_set_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gymrs_sharded_sync": (C.c_int, [C.c_void_p]),
    "gymrs_sharded_stats": (C.c_int, [C.c_void_p, f64p]),
    "gymrs_sharded_stats_clear": (C.c_int, [C.c_void_p]),
    "gymrs_sharded_reduce_path": (C.c_char_p, [C.c_void_p]),
    "gymrs_sharded_get_state": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "gymrs_sharded_get_step_result": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gymrs_last_error": (C.c_char_p, []),
    "gymrs_abi_version": (C.c_int, []),
}


def library_path() -> Path:
    return Path(os.environ.get("GYMRS_AMD_LIB", _PKG / "libgymrs_amd.so"))


def load_library() -> C.CDLL:
    """Load the HIP extension.  Raises if it has not been built (``python gym-rs_amd/build.py``)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise RuntimeError(
            f"gym-rs_amd: native library {path} is missing — build it with "
            "`python gym-rs_amd/build.py` (or __graft_entry__.build()). There is no CPU fallback."
        )
    try:
        # If the host application already loaded a HIP runtime (e.g. `import torch` on ROCm), the
        # dynamic linker resolves libamdhip64.so.7 to that same instance, so device pointers and
        # streams are interchangeable.
        lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    except OSError as exc:  # missing libamdhip64 etc.
        raise RuntimeError(f"gym-rs_amd: cannot load {path}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.gymrs_abi_version() != ABI_VERSION:
        raise RuntimeError("gym-rs_amd: ABI version mismatch between the binding and the library")
    _LIB = lib
    return lib
