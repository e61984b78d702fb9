"""Alias so that `import gymrs_amd` works: the package directory `gym-rs_amd/` (the name the build
contract asks for) contains a hyphen and cannot appear in an `import` statement."""
import importlib
import sys

_pkg = importlib.import_module("gym-rs_amd")
sys.modules[__name__] = _pkg
