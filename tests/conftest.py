import importlib
import json
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import spawn_server  # noqa: E402

spawn_server.start()  # before torch (hence the HIP runtime) is imported: the tests' child processes are forked by a process that never loads it

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "oversubscribed: starts several processes that share the one GPU of the test box (runs after every single-process correctness test)")
    config.addinivalue_line("markers", "perf: asserts a RATE or a timing relation (still runs under -m gpu, but after every correctness test)")


# The order of a `-m gpu -x` run (VERDICT r3 "next" #6, r5 "next" #3): parity against the oracle first, the machinery (twin / chain / fuzz) next, the
# in-process native sharder after that, then the bench contract -- and inside the non-perf block every test that starts SEVERAL processes on the one GPU
# (`oversubscribed` marker: a test mode, the likeliest to flake) LAST, so that a subprocess flake cannot hide product tests (round 5: one such failure cut
# off 23 tests, among them all of test_gpu_sharded_native).  Tests that assert a rate or a timing relation (`perf` marker) come after everything else.
GPU_ORDER = ["test_gpu_parity", "test_gpu_slowpaths", "test_gpu_envs", "test_gpu_pcg64_reset", "test_gpu_params_and_serde",
             "test_gpu_reset_log", "test_gpu_time_limit_elision", "test_gpu_rollout", "test_gpu_stats_readout", "test_gpu_aql_chain", "test_gpu_fuzz",
             "test_gpu_sharded_native", "test_gpu_bench_contract", "test_gpu_handover"]


def _order_key(indexed):
    index, item = indexed
    module = Path(str(item.fspath)).stem
    rank = GPU_ORDER.index(module) if module in GPU_ORDER else len(GPU_ORDER)
    block = 2 if item.get_closest_marker("perf") is not None else (1 if item.get_closest_marker("oversubscribed") is not None else 0)
    return (block, rank, index)


def pytest_collection_modifyitems(config, items):
    items[:] = [item for _, item in sorted(enumerate(items), key=_order_key)]
    # GPU tests never run by accident on a box without a device
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        # a test stuck inside a native call (a kernel that never ends, a host loop) must fail, not sit on the GPU box until
        # the runner's own limit: pytest-timeout's thread method dumps the stacks and exits even from inside ctypes
        if config.pluginmanager.hasplugin("timeout"):
            for item in items:
                if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                    item.add_marker(pytest.mark.timeout(600, method="thread"))
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_backtrace_on_fatal_signals(tmp_path_factory):
    """On a GPU box (or with GYMRS_TEST_SEGV_TRACE=1; =0 switches it off): print the native backtrace of the crashing thread on SIGSEGV & co
    (tests/cpp/segv_trace.c), then let python's faulthandler have its say.  Round 4: one GPU suite run in ~50 died with a segmentation fault
    while a test was starting a subprocess (a fork from a process whose HIP runtime was up); since then the tests' children are forked by a helper
    that never loads the runtime (spawn_server.py).  Round 5's soak (profiles/r05_suite_soak.log: 101 runs of the suite's product-heavy part with
    chains and with GYMRS_AQL=0, plus 30 full-suite runs) saw no crash under this handler; what it did find was a TEST race -- torch-initialised buffers
    handed to an engine's own non-blocking stream without waiting for torch's stream, 2 failures in 40 runs -- which is fixed."""
    import os

    want = os.environ.get("GYMRS_TEST_SEGV_TRACE")
    if want is None:
        try:
            import torch

            want = "1" if torch.cuda.is_available() else "0"
        except Exception:
            want = "0"
    if want != "1":
        yield
        return
    import ctypes

    out = tmp_path_factory.mktemp("segv") / "libsegv_trace.so"
    spawn_server.run(["gcc", "-O1", "-g", "-shared", "-fPIC", str(Path(__file__).resolve().parent / "cpp" / "segv_trace.c"), "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    assert lib.gymrs_test_install_segv_trace() == 0
    yield


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return json.loads((GOLDEN / f"{name}.json").read_text())

    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def twin():
    from oracle.bindings import Twin

    return Twin()


@pytest.fixture(scope="session")
def gymrs():
    try:
        import torch  # noqa: F401  (first, so the extension shares torch's HIP runtime instance)
    except Exception:
        pass
    return importlib.import_module("gym-rs_amd")
