"""drl_graph_exploration_amd — MI355X-native hot path of DRL_graph_exploration.

The belief step (simulate -> SLAM belief -> virtual-map covariance propagation -> utility / look-ahead
reward), graph export and the GCN policy forward/backward run as hand-written HIP kernels (gfx950)
behind the C ABI in include/drlgx.h.  This package is the host-side mirror of the reference's
Python surface (`ss2d`, `planner2d`, `scripts/Networks.py`, `scripts/policy.py`).

There is no CPU fallback: importing works anywhere (so the C-ABI symbols can be checked), but creating
an engine without a HIP device raises.
"""
from .config import DrlgxConfig, default_config, start_pose  # noqa: F401
from ._lib import lib, lib_path, DrlgxError  # noqa: F401

__all__ = ["DrlgxConfig", "default_config", "start_pose", "lib", "lib_path", "DrlgxError"]
