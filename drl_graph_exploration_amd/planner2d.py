"""`planner2d` — parameter / enum surface of the reference's second pybind module (src/Planner2D.cpp:9-106) that the DRL
scripts read. The planner itself (line planner, look-ahead reward, utility) runs in the engine: `drlgx_line_plan`,
`drlgx_lookahead`, `drlgx_utility`; the facade that binds them to the reference's call sites is
`pyplanner2d.EMExplorer`. The RRT / Dubins planners are outside the accelerated path (SURVEY.md §8 "out of scope")."""
import enum


class OptimizationAlgorithm(enum.IntEnum):
    """EMPlanner2D::OptimizationAlgorithm (include/em_exploration/Planner2D.h)."""
    EM_AOPT = 0
    EM_DOPT = 1
    OG_SHANNON = 2
    SLAM_OG_SHANNON = 3


class OptimizationResult(enum.IntEnum):
    SAMPLING_FAILURE = 0
    NO_SOLUTION = 1
    TERMINATION = 2
    SUCCESS = 3


class EMPlannerParameter(object):
    """EMPlanner2D::Parameter: the fields pyplanner2d.read_planner_params fills (scripts/envs/pyplanner2d.py:24-54)."""

    def __init__(self):
        self.verbose = False
        self.seed = 0
        self.max_edge_length = 2.0
        self.num_actions = 500
        self.max_nodes = 0.5
        self.angle_weight = 0.4
        self.distance_weight0 = 5.0
        self.distance_weight1 = 2.0
        self.d_weight = 0.0
        self.occupancy_threshold = 0.4
        self.safe_distance = 1.0
        self.alpha = 0.5
        self.reg_out = False
        self.algorithm = OptimizationAlgorithm.EM_AOPT
        self.dubins_control_model_enabled = False

    def pprint(self):
        print("EMPlanner Parameters", vars(self))


class EMPlanner2D(object):
    """Namespace holder so that `planner2d.EMPlanner2D.OptimizationAlgorithm.EM_AOPT` resolves as in the reference."""
    OptimizationAlgorithm = OptimizationAlgorithm
    OptimizationResult = OptimizationResult
