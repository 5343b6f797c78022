"""Import alias for the reference's wrappers: `import build.ss2d as ss2d` (scripts/envs/pyss2d.py:7) and
`import build.planner2d as planner2d` (scripts/envs/pyplanner2d.py:6) resolve to the drlgx-backed modules when this
repository's root is on sys.path (the reference puts its cmake `build/` directory there)."""
import sys

from drl_graph_exploration_amd import planner2d, ss2d

sys.modules[__name__ + ".ss2d"] = ss2d
sys.modules[__name__ + ".planner2d"] = planner2d
