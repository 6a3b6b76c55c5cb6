"""Synthetic maps live in the package (synth.py); tests import them through this alias."""
import vhp_amd  # noqa: F401
from importlib import import_module

_s = import_module("visibility-heuristic-path-planner_amd.synth")
globals().update({k: getattr(_s, k) for k in dir(_s) if not k.startswith("__")})
