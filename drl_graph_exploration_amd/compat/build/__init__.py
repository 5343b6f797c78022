"""Import alias for the reference's wrappers: `import build.ss2d as ss2d` (scripts/envs/pyss2d.py:7) and
`import build.planner2d as planner2d` (scripts/envs/pyplanner2d.py:6) resolve to the drlgx-backed modules once
`drl_graph_exploration_amd/compat` is on sys.path - `drl_graph_exploration_amd.compat.enable()` puts it there (the reference puts its
cmake `build/` directory's parent there).  Kept out of the repository root: a top-level package named `build` shadows the PyPA
`build` module (`python -m build`) and sits where setuptools / cmake put their output."""
import sys

from drl_graph_exploration_amd import planner2d, ss2d

sys.modules[__name__ + ".ss2d"] = ss2d
sys.modules[__name__ + ".planner2d"] = planner2d
