"""Single-environment facades `SS2D` / `EMExplorer` with the reference's Python-wrapper surface
(scripts/envs/pyss2d.py:56-330, scripts/envs/pyplanner2d.py:57-84) over a one-env drlgx engine.

They exist so that code written against the reference wrappers (`ExplorationEnv`, notebooks, `test.py` reporting) can
run on the HIP path one environment at a time; throughput work should use `VecExplorationEnv`. The members the
reference's `ExplorationEnv` touches are provided:

    sim.simulate(odom)  sim.calculate_utility(d)  sim.line_plan(goal_key, fro)  sim.simulations_reward(actions)
    sim.step  sim.vehicle_position  sim.environment.get_landmark(key)
    sim._slam.key_size() / adjacency_degree_get() / adjacency_out() / features_out() / get_key_points(i)
    sim._slam.map.get_landmark_size() / iter_landmarks() / iter_trajectory() / get_current_vehicle()
    sim._virtual_map.to_array() / to_cov_trace() / explored() / get_parameter()
    sim._planner_params / _environment_params / _map_params / _virtual_map_params / _sensor_params / _control_params

`move / measure / optimize / update_virtual_map` are one fused device step and are not exposed separately.
"""
import math
from configparser import ConfigParser

import numpy as np
import torch

from . import planner2d, ss2d
from .config import DrlgxConfig, start_pose
from .engine import Engine


def load_config(path):
    cp = ConfigParser(inline_comment_prefixes=(";",))
    cp.read(path)
    return cp


def config_from_ini(cp, max_poses=256, max_actions=None, max_snapshots=1):
    """The ini sections of pyss2d.py:10-55 + pyplanner2d.py:24-54 -> struct drlgx_config (plus the parameter objects)."""
    sp = ss2d.BearingRangeSensorModelParameter()
    sp.bearing_noise = math.radians(cp.getfloat("Sensor Model", "bearing_noise"))
    sp.range_noise = cp.getfloat("Sensor Model", "range_noise")
    sp.min_bearing = math.radians(cp.getfloat("Sensor Model", "min_bearing"))
    sp.max_bearing = math.radians(cp.getfloat("Sensor Model", "max_bearing"))
    sp.min_range = cp.getfloat("Sensor Model", "min_range")
    sp.max_range = cp.getfloat("Sensor Model", "max_range")
    cm = ss2d.SimpleControlModelParameter()
    cm.rotation_noise = math.radians(cp.getfloat("Control Model", "rotation_noise"))
    cm.translation_noise = cp.getfloat("Control Model", "translation_noise")
    ep = ss2d.EnvironmentParameter()
    for k in ("min_x", "max_x", "min_y", "max_y", "max_steps", "safe_distance"):
        setattr(ep, k, cp.getfloat("Environment", k))
    mp = ss2d.EnvironmentParameter()
    ext = 20.0  # read_map_params(config, ext=20.0)
    mp.min_x, mp.max_x, mp.min_y, mp.max_y = ep.min_x - ext, ep.max_x + ext, ep.min_y - ext, ep.max_y + ext
    mp.safe_distance = ep.safe_distance
    vp = ss2d.VirtualMapParameter(mp)
    vp.resolution = cp.getfloat("Virtual Map", "resolution")
    vp.sigma0 = cp.getfloat("Virtual Map", "sigma0")
    vp.num_samples = cp.getint("Virtual Map", "num_samples")
    pp = planner2d.EMPlannerParameter()
    if cp.has_section("Planner"):
        for k in ("angle_weight", "distance_weight0", "distance_weight1", "d_weight", "max_edge_length", "occupancy_threshold",
                  "safe_distance", "max_nodes"):
            if cp.has_option("Planner", k):
                setattr(pp, k, cp.getfloat("Planner", k))
        pp.seed = cp.getint("Planner", "seed")
        pp.reg_out = cp.getboolean("Planner", "reg_out")
        pp.algorithm = planner2d.OptimizationAlgorithm[cp.get("Planner", "algorithm")]
    c = DrlgxConfig()
    c.bearing_noise, c.range_noise = sp.bearing_noise, sp.range_noise
    c.min_bearing, c.max_bearing, c.min_range, c.max_range = sp.min_bearing, sp.max_bearing, sp.min_range, sp.max_range
    c.translation_noise, c.rotation_noise = cm.translation_noise, cm.rotation_noise
    c.env_min_x, c.env_max_x, c.env_min_y, c.env_max_y, c.safe_distance = ep.min_x, ep.max_x, ep.min_y, ep.max_y, ep.safe_distance
    c.map_min_x, c.map_max_x, c.map_min_y, c.map_max_y = mp.min_x, mp.max_x, mp.min_y, mp.max_y
    c.resolution, c.sigma0, c.num_samples = vp.resolution, vp.sigma0, vp.num_samples
    c.sigma_x0 = cp.getfloat("Simulator", "sigma_x0")
    c.sigma_y0 = cp.getfloat("Simulator", "sigma_y0")
    c.sigma_theta0 = math.radians(cp.getfloat("Simulator", "sigma_theta0"))
    fixed = []
    if cp.has_section("Landmarks") and cp.has_option("Landmarks", "x") and cp.has_option("Landmarks", "y"):  # pyss2d.py:107-115
        import ast
        fixed = list(zip(ast.literal_eval(cp.get("Landmarks", "x")), ast.literal_eval(cp.get("Landmarks", "y"))))
    c.num_landmarks = cp.getint("Simulator", "num") + len(fixed)
    c.angle_weight, c.distance_weight0, c.distance_weight1 = pp.angle_weight, pp.distance_weight0, pp.distance_weight1
    c.occupancy_threshold, c.max_edge_length, c.algorithm = pp.occupancy_threshold, pp.max_edge_length, int(pp.algorithm)
    c.max_poses = max_poses
    c.max_landmarks = max(1, min(c.num_landmarks, 127))
    c.max_factors = max(64, 12 * max_poses)
    if max_actions is None:  # the longest line plan the boxes allow (config.default_config)
        span = max(ep.max_x - ep.min_x, ep.max_y - ep.min_y, (mp.max_x - mp.min_x) / 2, (mp.max_y - mp.min_y) / 2)
        max_actions = int(math.ceil(math.hypot(span, span) / c.max_edge_length)) + 3
    c.max_actions, c.max_snapshots = max_actions, max_snapshots
    return c, dict(sensor=sp, control=cm, environment=ep, map=mp, virtual_map=vp, planner=pp, fixed_landmarks=fixed)


class _Map(object):
    """SLAM2D.map / Simulator2D.environment: an `Environment` view (src/SS2D.cpp:141-171)."""

    def __init__(self, owner, truth):
        self._o, self._truth = owner, truth

    def get_landmark_size(self):
        return self._o.engine.cfg.num_landmarks if self._truth else self._o.engine.counts(0)["landmarks"]

    def get_trajectory_size(self):
        return self._o.engine.counts(0)["poses"]

    @property
    def distance(self):
        """Environment::getDistance (Simulator2D.cpp:244-250)."""
        return ss2d.trajectory_distance([v.pose for v in self.iter_trajectory()])

    def iter_landmarks(self):
        if self._truth:
            _, lms = self._o.engine.ground_truth(0)
            for k, p in enumerate(lms):
                yield k, ss2d.LandmarkBeliefState(ss2d.Point2(*p))
        else:
            keys, xy, info = self._o.engine.landmarks(0)
            for k, p, i in zip(keys, xy, info):
                yield int(k), ss2d.LandmarkBeliefState(ss2d.Point2(*p), i)

    def get_landmark(self, key):
        for k, l in self.iter_landmarks():
            if k == key:
                return l
        raise KeyError(key)

    def iter_trajectory(self):
        if self._truth:
            veh, _ = self._o.engine.ground_truth(0)
            yield ss2d.VehicleBeliefState(ss2d.Pose2(*veh))
            return
        xyt, info = self._o.engine.poses(0)
        for p, i in zip(xyt, info):
            yield ss2d.VehicleBeliefState(ss2d.Pose2(*p), i)

    def get_current_vehicle(self):
        if self._truth:
            return next(self.iter_trajectory())
        xyt, info = self._o.engine.poses(0)
        return ss2d.VehicleBeliefState(ss2d.Pose2(*xyt[-1]), info[-1])


class _Slam(object):
    """SLAM2D getters used by graph_matrix (src/SLAM2D.cpp:141-273)."""

    def __init__(self, owner):
        self._o = owner
        self.map = _Map(owner, False)

    def key_size(self):
        c = self._o.engine.counts(0)
        return c["poses"] + c["landmarks"]

    def adjacency_degree_get(self):
        self._A, self._X = self._o.engine.adjacency(0)

    def adjacency_out(self):
        return self._A

    def features_out(self):
        return self._X.reshape(-1, 1)

    def get_key_points(self, i):
        keys, lxy, _ = self._o.engine.landmarks(0)
        if i < len(keys):
            return [lxy[i][0], lxy[i][1]]
        xyt, _ = self._o.engine.poses(0)
        return [xyt[i - len(keys)][0], xyt[i - len(keys)][1]]


class _VirtualMap(object):
    def __init__(self, owner):
        self._o = owner

    def to_array(self):
        return self._o.engine.virtual_map(0)[0]

    def to_cov_trace(self):
        return self._o.engine.virtual_map(0)[2]

    def to_cov_array(self):
        ln, an = self._o.engine.cov_array()
        return ln[0].cpu().numpy(), an[0].cpu().numpy()

    def explored(self):
        return float(self._o.engine.explored()[0])

    def get_parameter(self):
        return self._o._virtual_map_params


class _Sim(object):
    def __init__(self, owner):
        self._o = owner
        self.environment = _Map(owner, True)

    @property
    def vehicle(self):
        return ss2d.Pose2(*self._o.engine.ground_truth(0)[0])


class SS2D(object):
    """scripts/envs/pyss2d.py:56-330 (construction + simulate + getters).

    One fused device step per `simulate` (`drlgx_step`).  (The same simulation driven call by call through the `ss2d`
    module's own classes, in the reference's order, lives in tests/staged_facade.py - the module classes are the API, that
    driver is test code.)"""

    def __init__(self, config, verbose=False, device=0, max_poses=256, start=None):
        self._config = load_config(config) if isinstance(config, str) else config
        cfg, prm = config_from_ini(self._config, max_poses=max_poses)
        self._sensor_params, self._control_params = prm["sensor"], prm["control"]
        self._environment_params, self._map_params = prm["environment"], prm["map"]
        self._virtual_map_params, self._planner_params = prm["virtual_map"], prm["planner"]
        lo = int(self._config.getfloat("Simulator", "lo"))
        seed = self._config.getint("Simulator", "seed")
        # `start` (x, y, theta) overrides the reference's integer start pose (extension used by tests)
        x0, y0, theta0 = start_pose(lo, cfg.map_max_x) if start is None else start
        self.verbose = verbose
        self.engine = Engine(cfg, 1, max(cfg.max_landmarks, 1), device)
        if prm["fixed_landmarks"]:
            self.engine.set_fixed_landmarks(prm["fixed_landmarks"])
        self.engine.reset([0], [seed], starts=np.array([[x0, y0, theta0]]))
        self.engine.check_status()
        self._slam, self._virtual_map, self._sim = _Slam(self), _VirtualMap(self), _Sim(self)

    @property
    def step(self):
        return self.engine.counts(0)["step"]

    def simulate(self, odom, core=True):
        """pyss2d.py:171-206; returns True when the odometry is rejected (outside the map box)."""
        if not core:
            raise NotImplementedError("non-core (Dubins intermediate) steps are outside the accelerated path")
        mp = self._map_params
        if not mp.min_x < odom[0] < mp.max_x or not mp.min_y < odom[1] < mp.max_y:
            return True
        self.engine.step(torch.tensor([[odom[0], odom[1], odom[2]]], dtype=torch.float64, device=self.engine.device))
        self.engine.check_status()
        return False

    @property
    def vehicle_position(self):
        return self._slam.map.get_current_vehicle().pose

    @property
    def environment(self):
        return self._sim.environment

    @property
    def map(self):
        return self._slam.map

    @property
    def distance(self):
        """pyss2d.py:216-218: the travelled distance of the estimated trajectory."""
        return self._slam.map.distance


class EMExplorer(SS2D):
    """scripts/envs/pyplanner2d.py:57-84."""

    def calculate_utility(self, distance):
        d = torch.tensor([float(distance)], dtype=torch.float64, device=self.engine.device)
        return float(self.engine.utility(d)[0])

    def line_plan(self, goal_key, fron):
        ce = torch.zeros(1, dtype=torch.int32, device=self.engine.device)
        goal = torch.tensor([[fron[0], fron[1]]], dtype=torch.float64, device=self.engine.device)
        acts, n = self.engine.line_plan(ce, goal)
        self._last_plan = (acts, n)
        return [ss2d.Pose2(*a) for a in acts[0, :int(n[0])].cpu().numpy()]

    def simulations_reward(self, actions):
        A = self.engine.cfg.max_actions
        acts = torch.zeros(1, A, 3, dtype=torch.float64, device=self.engine.device)
        for k, a in enumerate(actions):
            acts[0, k] = torch.tensor([a.x, a.y, a.theta] if hasattr(a, "x") else list(a), dtype=torch.float64)
        n = torch.tensor([len(actions)], dtype=torch.int32, device=self.engine.device)
        ce = torch.zeros(1, dtype=torch.int32, device=self.engine.device)
        return float(self.engine.lookahead(ce, acts, n)[0])
