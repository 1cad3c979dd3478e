"""fit-slam_amd — MI355X-native frontier scoring (ray-cast arrival information + landmark Fisher
information) behind the C ABI of include/fitslam_frontier.h.

The directory name carries a hyphen (project naming); import it with
    importlib.import_module("fit-slam_amd")
The package holds only what the hot path needs: csrc/ (HIP kernels + C ABI), capi.py (ctypes
binding), synth.py (synthetic workloads), shard.py (multi-GPU sharding) and host/ (C++ mirror of the
reference interface).  It never imports oracle/.
"""
from . import _build, capi, synth  # noqa: F401
from .capi import FrontierScorer, FsError, MultiScorer, RECORD_DTYPE, load_library  # noqa: F401

__all__ = ["FrontierScorer", "MultiScorer", "FsError", "RECORD_DTYPE", "load_library", "capi", "synth", "_build"]
