"""Compatibility shims for scripts written against the reference's layout."""
import os
import sys


def enable():
    """Make the reference's import lines (`import build.ss2d as ss2d`, `import build.planner2d as planner2d`) resolve to the
    drlgx-backed modules: this directory goes to the FRONT of sys.path (it holds the `build` alias package)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return here
