"""Import shim: the package directory `visibility-heuristic-path-planner_amd` is not a Python identifier."""
import importlib
import sys

_pkg = importlib.import_module("visibility-heuristic-path-planner_amd")
sys.modules[__name__] = _pkg
