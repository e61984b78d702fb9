import importlib
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
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
