"""`planner2d` — the module surface of the reference's second pybind module (src/Planner2D.cpp:9-106) that the DRL
scripts use: the parameter classes and enums, and `EMPlanner2D(parameter, sensor_model, control_model)` with
`calculate_utility` / `line_planner` / `simulations_reward` over the engine of the simulation the models belong to
(`ss2d.Simulator2D`).  The RRT / EM / Dubins planners are outside the accelerated path (SURVEY.md §8 "out of scope")."""
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

        self.dubins_parameter = None

    def pprint(self):
        print("EMPlanner Parameters", vars(self))


class DubinsParameter(object):
    """EMPlanner2D::DubinsParameter (src/Planner2D.cpp:12-22): carried for the ini reader; the Dubins path library is
    outside the accelerated path."""

    def __init__(self):
        self.max_w = self.dw = self.min_v = self.max_v = self.dv = self.dt = 0.0
        self.min_duration = self.max_duration = self.tolerance_radius = 0.0


class EMPlanner2D(object):
    """`EMPlanner2D(parameter, sensor_model, control_model)` (src/Planner2D.cpp:73-91) over the engine that the models'
    simulation owns: the calls the DRL scripts make - `calculate_utility` (static), `line_planner`, `simulations_reward` -
    run `drlgx_utility` / `drlgx_line_plan` / `drlgx_lookahead`.  The sampling planners (`optimize`, `optimize2`,
    `rrt_planner`, the Dubins library) are outside the accelerated path (out of scope)."""
    OptimizationAlgorithm = OptimizationAlgorithm
    OptimizationResult = OptimizationResult

    def __init__(self, parameter, sensor_model, control_model):
        self._ses = sensor_model._session
        if self._ses is None or control_model._session is not self._ses:
            raise ValueError("the sensor and control models must come from one Simulator2D")
        self.set_parameter(parameter)

    def get_parameter(self):
        return self._parameter

    def set_parameter(self, parameter):
        self._parameter = parameter
        self._ses.planner_params = parameter
        if self._ses.engine is not None:  # (constructed after SLAM2D.add_prior, as pyplanner2d.EMExplorer does)
            self._ses.engine.set_planner_parameter(parameter.angle_weight, parameter.distance_weight0, parameter.distance_weight1,
                                                   parameter.occupancy_threshold, parameter.max_edge_length, int(parameter.algorithm))

    @staticmethod
    def calculate_utility(virtual_map, distance, parameter):
        """EMPlanner2D::calculateUtility (Planner2D.cpp:354-366)."""
        import torch
        e = virtual_map._ses.require_engine()
        c = e.cfg
        if (parameter.distance_weight0, parameter.distance_weight1, parameter.occupancy_threshold) != \
                (c.distance_weight0, c.distance_weight1, c.occupancy_threshold):
            e.set_planner_parameter(parameter.angle_weight, parameter.distance_weight0, parameter.distance_weight1,
                                    parameter.occupancy_threshold, parameter.max_edge_length, int(parameter.algorithm))
        return float(e.utility(torch.tensor([float(distance)], dtype=torch.float64, device=e.device))[0])

    def line_planner(self, slam, virtual_map, n_key, fron_0, fron_1):
        """EMPlanner2D::line_planner (Planner2D.cpp:937-1041): goal = node `n_key` of the pose graph if it is one, else the
        frontier point (fron_0, fron_1); returns the list of Pose2 odometry actions."""
        import torch
        from . import ss2d
        e = self._ses.require_engine()
        if n_key < slam.key_size():
            fron_0, fron_1 = slam.get_key_points(n_key)
        ce = torch.zeros(1, dtype=torch.int32, device=e.device)
        goal = torch.tensor([[fron_0, fron_1]], dtype=torch.float64, device=e.device)
        acts, n = e.line_plan(ce, goal)
        e.check_status()
        return [ss2d.Pose2(*a) for a in acts[0, :int(n[0])].cpu().numpy()]

    def simulations_reward(self, slam, virtual_map, simulator, actions):
        """EMPlanner2D::simulations_reward (Planner2D.cpp:1416-1468): look-ahead with copies of the SLAM, map and
        simulator state (including its RNG streams); the live objects are not modified."""
        import torch
        e = self._ses.require_engine()
        A = e.cfg.max_actions
        if len(actions) > A:
            raise ValueError("plan longer than the engine's max_actions")
        acts = torch.zeros(1, A, 3, dtype=torch.float64)
        for k, a in enumerate(actions):
            acts[0, k] = torch.tensor([a.x, a.y, a.theta] if hasattr(a, "x") else list(a), dtype=torch.float64)
        n = torch.tensor([len(actions)], dtype=torch.int32, device=e.device)
        ce = torch.zeros(1, dtype=torch.int32, device=e.device)
        r = float(e.lookahead(ce, acts.to(e.device), n)[0])
        e.check_status()
        return r

    def _out_of_scope(self, *a, **k):
        raise NotImplementedError("sampling planners (EM / RRT / Dubins) are outside the accelerated path")

    optimize = optimize2 = rrt_planner = iter_solution = iter_rrt = get_dubins_path = _out_of_scope
