#!/usr/bin/env python
"""Developer builds for kernel experiments (not part of the product): a library variant with extra -D flags and only
the headline instantiations of the step-kernel table (GYMRS_DEV_MINIMAL), written to _ab/lib<name>.so in ~1 minute.

    python tools/devbuild.py NAME [-DMACRO[=V] ...]      # then: python tools/step_timer.py --lib _ab/libNAME.so --lib ...
"""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("gymrs_amd_build", ROOT / "gym-rs_amd" / "build.py")
build = importlib.util.module_from_spec(spec)
spec.loader.exec_module(build)
name = sys.argv[1]
flags = ["-DGYMRS_DEV_MINIMAL"] + sys.argv[2:]
out = ROOT / "_ab" / f"lib{name}.so"
out.parent.mkdir(exist_ok=True)
build.build_hip(force=True, extra_flags=flags, out=out)
print(out)
