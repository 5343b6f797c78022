"""`ss2d` — the module surface of the reference's first pybind module (src/SS2D.cpp:18-258) over the drlgx engine.

Value types (Pose2 / Point2 / Rot2 with gtsam semantics), the parameter classes with the reference's property names, and
the classes with behaviour the reference's Python layer constructs and drives (scripts/envs/pyss2d.py:59-246):
`Simulator2D`, `SLAM2D`, `VirtualMap`, `Environment` (+ the sensor / control model holders).  They are thin objects over
ONE one-environment engine: every call of the reference's call sequence maps onto one staged C-ABI call
(include/drlgx.h: drlgx_stage_*), e.g.

    sim.move(odom, True) + slam.add_odometry(cs)         -> drlgx_stage_move
    sim.measure()                                         -> drlgx_stage_measure          (values returned, nothing added)
    slam.add_measurement(key, m) ... slam.optimize()      -> drlgx_stage_add_measurements + drlgx_stage_optimize
    vm.update_probability(slam, sensor) / update_information(map, sensor) -> drlgx_stage_update_map

The objects of one simulation find each other through the values the reference's script passes between them: the
`Pose2` of `Simulator2D.vehicle` carries its simulation into `SLAM2D.add_prior`, the `ControlState` of `Simulator2D.move`
into `add_odometry`, `slam` / `sensor_model` into `VirtualMap.update_*`; every such call checks that both sides belong to
ONE simulation and raises otherwise.  A `SLAM2D` / `VirtualMap` constructed without `simulator=` joins the most recently
constructed `Simulator2D` that still lacks one (the construction order of pyss2d.py:105-108); a second one for the same
simulator raises.  The engine is created at `SLAM2D.add_prior` (the first call that needs every parameter).  The batched product path does not go through these classes (`VecExplorationEnv`); they exist so
that code written against the reference's module keeps working, one environment at a time.  No CPU path: creating the
engine without a HIP device raises.
"""
import math

import numpy as np


def _wrap(th):  # Rot2(th).theta()
    return math.atan2(math.sin(th), math.cos(th))


class Rot2(object):
    def __init__(self, theta=0.0):
        self._c, self._s = math.cos(theta), math.sin(theta)

    @property
    def theta(self):
        return math.atan2(self._s, self._c)

    def __repr__(self):
        return "[%f]" % self.theta


class Point2(object):
    def __init__(self, x=0.0, y=0.0):
        self.x, self.y = float(x), float(y)

    def __repr__(self):
        return "[%f, %f]" % (self.x, self.y)


class Pose2(object):
    """gtsam::Pose2 (x, y, cos, sin): `theta` = atan2(sin, cos); `*` = compose (src/SS2D.cpp:21-36)."""

    def __init__(self, x=0.0, y=0.0, theta=0.0):
        self.x, self.y = float(x), float(y)
        self._c, self._s = math.cos(theta), math.sin(theta)
        self._theta_in = float(theta)  # (kept so that a start pose reaches the engine with the angle it was given)

    @property
    def theta(self):
        return math.atan2(self._s, self._c)

    def __mul__(self, o):
        r = Pose2()
        del r._theta_in
        c, s = self._c * o._c - self._s * o._s, self._s * o._c + self._c * o._s
        if abs(c * c + s * s - 1.0) > 1e-9:
            n = math.sqrt(c * c + s * s)
            c, s = c / n, s / n
        r._c, r._s = c, s
        r.x = self.x + (self._c * o.x - self._s * o.y)
        r.y = self.y + (self._s * o.x + self._c * o.y)
        return r

    def __repr__(self):
        return "[%f, %f, %f]" % (self.x, self.y, self.theta)


class BearingRangeSensorModelParameter(object):
    """BearingRangeSensorModel::Parameter (include/em_exploration/Simulation2D.h:47-76); angle setters wrap."""

    def __init__(self):
        self._bn = self._minb = self._maxb = 0.0
        self.range_noise = self.min_range = self.max_range = 0.0

    bearing_noise = property(lambda s: s._bn, lambda s, v: setattr(s, "_bn", _wrap(v)))
    min_bearing = property(lambda s: s._minb, lambda s, v: setattr(s, "_minb", _wrap(v)))
    max_bearing = property(lambda s: s._maxb, lambda s, v: setattr(s, "_maxb", _wrap(v)))

    def pprint(self):
        print("BearingRangeSensorModel Parameters", vars(self))


class SimpleControlModelParameter(object):
    def __init__(self):
        self.translation_noise = 0.0
        self._rn = 0.0

    rotation_noise = property(lambda s: s._rn, lambda s, v: setattr(s, "_rn", _wrap(v)))

    def pprint(self):
        print("SimpleControlModel Parameters", vars(self))


class EnvironmentParameter(object):
    def __init__(self):
        self.min_x = self.max_x = self.min_y = self.max_y = 0.0
        self.max_steps = 0.0
        self.safe_distance = 0.0

    def pprint(self):
        print("Environment Parameters", vars(self))


class VirtualMapParameter(EnvironmentParameter):
    def __init__(self, env_param):
        super().__init__()
        self.__dict__.update(env_param.__dict__)
        self.resolution, self.sigma0, self.num_samples = 2.0, 2.0, 20  # VirtualMap.cpp:7-8 defaults

    def pprint(self):
        print("Virtual Map Parameters", vars(self))


class VehicleBeliefState(object):
    def __init__(self, pose=None, information=None):
        self.pose = pose if pose is not None else Pose2()
        self.information = np.eye(3) if information is None else np.asarray(information, dtype=np.float64)
        self.core_vehicle = True

    @property
    def covariance(self):
        return np.linalg.inv(self.information)


class LandmarkBeliefState(object):
    def __init__(self, point=None, information=None):
        self.point = point if point is not None else Point2()
        self.information = np.eye(2) if information is None else np.asarray(information, dtype=np.float64)

    @property
    def covariance(self):
        return np.linalg.inv(self.information)


class Measurement(object):
    """BearingRangeSensorModel::Measurement (bearing, range[, sigmas])."""

    def __init__(self, bearing, range_, sigmas=None):
        self.bearing, self.range = bearing, range_
        self.sigmas = None if sigmas is None else np.asarray(sigmas, dtype=np.float64)
        self.has_jacobian = False

    def transform_from(self, origin):
        q = Point2(self.range * math.cos(self.bearing), self.range * math.sin(self.bearing))
        return Point2(origin._c * q.x - origin._s * q.y + origin.x, origin._s * q.x + origin._c * q.y + origin.y)


def trajectory_distance(poses, angle_weight=0.5):
    d = 0.0
    for a, b in zip(poses[:-1], poses[1:]):
        dx, dy = b.x - a.x, b.y - a.y
        bearing = math.atan2(-a._s * dx + a._c * dy, a._c * dx + a._s * dy)  # Pose2::bearing(Pose2).theta()
        d += math.sqrt(math.hypot(dx, dy) ** 2 + (bearing * angle_weight) ** 2)
    return d


# ------------------------------------------------------------------------------------------------------------------
# classes with behaviour (engine-backed)
# ------------------------------------------------------------------------------------------------------------------
class _Session(object):
    """The objects of one simulation + their engine (created lazily at SLAM2D.add_prior)."""
    _latest = None  # the most recently constructed Simulator2D's session: only a DEFAULT for `simulator=None`

    def __init__(self):
        self.sim = self.slam = self.vm = self.engine = None
        self.planner_params = None
        self.max_poses = 256

    @staticmethod
    def join(role, obj, simulator):
        """Attach `obj` as the `role` ('slam' / 'vm') of `simulator`'s session (default: the latest simulator).  One object
        per role: a second one raises instead of silently re-wiring another simulation's parts."""
        ses = simulator._ses if simulator is not None else _Session._latest
        if ses is None:
            raise RuntimeError("construct a Simulator2D first (or pass simulator=...)")
        if getattr(ses, role) is not None:
            raise RuntimeError("this Simulator2D already has its %s: simulations constructed interleaved must be wired "
                               "explicitly, %s(..., simulator=sim)" % (role, type(obj).__name__))
        setattr(ses, role, obj)
        return ses

    def require_engine(self):
        if self.engine is None:
            raise RuntimeError("the engine of this simulation is created by SLAM2D.add_prior (call order of pyss2d.SS2D.__init__)")
        return self.engine

    def materialise(self, prior_state):
        import torch  # noqa: F401  (device memory)
        from .config import DrlgxConfig
        from .engine import Engine
        from . import planner2d
        sim, slam, vm = self.sim, self.slam, self.vm
        if sim is None or sim._env_params is None:
            raise RuntimeError("Simulator2D.random_landmarks must be called before SLAM2D.add_prior")
        sp, cp, ep, mp = sim._sensor_params, sim._control_params, sim._env_params, slam._map_params
        vp = vm._params if vm is not None else VirtualMapParameter(mp)
        pp = self.planner_params or planner2d.EMPlannerParameter()
        c = DrlgxConfig()
        c.bearing_noise, c.range_noise = sp.bearing_noise, sp.range_noise
        c.min_bearing, c.max_bearing, c.min_range, c.max_range = sp.min_bearing, sp.max_bearing, sp.min_range, sp.max_range
        c.translation_noise, c.rotation_noise = cp.translation_noise, cp.rotation_noise
        c.env_min_x, c.env_max_x, c.env_min_y, c.env_max_y, c.safe_distance = ep.min_x, ep.max_x, ep.min_y, ep.max_y, ep.safe_distance
        c.map_min_x, c.map_max_x, c.map_min_y, c.map_max_y = mp.min_x, mp.max_x, mp.min_y, mp.max_y
        c.resolution, c.sigma0, c.num_samples = vp.resolution, vp.sigma0, vp.num_samples
        info = np.asarray(prior_state.information, dtype=np.float64)
        if info.shape != (3, 3) or np.abs(info - info.T).max() > 0 or not (np.diag(info) > 0).all():
            raise ValueError("the prior information must be a symmetric 3 x 3 matrix with a positive diagonal")
        full_prior = np.abs(info - np.diag(np.diag(info))).max() > 0  # (installed after the staged reset, below)
        c.sigma_x0, c.sigma_y0, c.sigma_theta0 = (1.0 / math.sqrt(info[k, k]) for k in range(3))
        c.num_landmarks = sim._num_landmarks
        c.angle_weight, c.distance_weight0, c.distance_weight1 = pp.angle_weight, pp.distance_weight0, pp.distance_weight1
        c.occupancy_threshold, c.max_edge_length, c.algorithm = pp.occupancy_threshold, pp.max_edge_length, int(pp.algorithm)
        c.max_poses = self.max_poses
        c.max_landmarks = max(1, min(c.num_landmarks, 127))
        c.max_factors = max(64, 12 * c.max_poses)
        span = max(ep.max_x - ep.min_x, ep.max_y - ep.min_y, (mp.max_x - mp.min_x) / 2, (mp.max_y - mp.min_y) / 2)
        c.max_actions = int(math.ceil(math.hypot(span, span) / c.max_edge_length)) + 3
        c.max_snapshots = 1
        self.engine = Engine(c, 1, max(c.max_landmarks, 1), sim._device)
        p, v = prior_state.pose, sim._start
        # (the reference's own caller passes the vehicle's pose, pyss2d.py:124-135; any other pose becomes the prior factor's pose and
        # the initial estimate of x0, SLAM2D.cpp:44-57, while the simulator keeps its vehicle where it is)
        own_pose = (abs(p.x - v.x), abs(p.y - v.y)) != (0.0, 0.0) or abs(_wrap(p.theta - v.theta)) > 1e-15
        if sim._fixed_landmarks:
            self.engine.set_fixed_landmarks(sim._fixed_landmarks)
        self.engine.stage_reset([0], [sim._seed], np.array([[v.x, v.y, getattr(v, "_theta_in", v.theta)]]))
        if own_pose:
            self.engine.stage_set_prior_pose(0, (p.x, p.y, getattr(p, "_theta_in", p.theta)))
        if full_prior:
            self.engine.stage_set_prior_information(0, info)
        self.engine.stage_update_map(rebuild=False)  # sums of the untouched map (a fresh VirtualMap)
        self.engine.check_status()


class BearingRangeSensorModel(object):
    """Holder handed from `Simulator2D.sensor_model` to `VirtualMap.update_*` / `EMPlanner2D` (src/SS2D.cpp:83-88)."""

    def __init__(self, parameter, seed=0, _session=None):
        self.parameter, self._session = parameter, _session


class SimpleControlModel(object):
    def __init__(self, parameter, seed=0, _session=None):
        self.parameter, self._session = parameter, _session


class SimpleControlModelState(object):
    """SimpleControlModel::ControlState (src/SS2D.cpp:98-106): what `Simulator2D.move` returns and `SLAM2D.add_odometry` takes."""

    def __init__(self, pose=None, odom=None, sigmas=None):
        self.pose, self.odom = pose or Pose2(), odom or Pose2()
        self.sigmas = np.zeros(3) if sigmas is None else np.asarray(sigmas, dtype=np.float64)
        self.has_jacobian = False


class Environment(object):
    """`Simulator2D.environment` (ground truth) / `SLAM2D.map` (estimates): the getters of src/SS2D.cpp:141-171."""

    def __init__(self, session, truth, parameter=None):
        self._ses, self._truth, self.parameter = session, truth, parameter

    @property
    def distance(self):
        """Environment::getDistance (Simulator2D.cpp:244-250): path length of the trajectory with the heading term of
        `sqDistanceBetweenPoses(a, b, 0.5)` (Distance.cpp:5-9: range^2 + (0.5 * bearing of b seen from a)^2)."""
        return trajectory_distance([v.pose for v in self.iter_trajectory()])

    def get_landmark_size(self):
        e = self._ses.require_engine()
        return e.cfg.num_landmarks if self._truth else e.counts(0)["landmarks"]

    def get_trajectory_size(self):
        return 1 if self._truth else self._ses.require_engine().counts(0)["poses"]

    def iter_landmarks(self):
        e = self._ses.require_engine()
        if self._truth:
            _, lms = e.ground_truth(0)
            for k, p in enumerate(lms):
                yield k, LandmarkBeliefState(Point2(*p))
        else:
            keys, xy, info = e.landmarks(0)
            for k, p, i in zip(keys, xy, info):
                yield int(k), LandmarkBeliefState(Point2(*p), i)

    def get_landmark(self, key):
        for k, l in self.iter_landmarks():
            if k == key:
                return l
        raise KeyError(key)

    def iter_trajectory(self):
        e = self._ses.require_engine()
        if self._truth:
            veh, _ = e.ground_truth(0)
            yield VehicleBeliefState(Pose2(*veh))
            return
        xyt, info = e.poses(0)
        for p, i in zip(xyt, info):
            yield VehicleBeliefState(Pose2(*p), i)

    def get_vehicle(self, i):
        return list(self.iter_trajectory())[i]

    def get_current_vehicle(self):
        e = self._ses.require_engine()
        if self._truth:
            return next(self.iter_trajectory())
        xyt, info = e.poses(0)
        return VehicleBeliefState(Pose2(*xyt[-1]), info[-1])


class Simulator2D(object):
    """src/SS2D.cpp:173-187.  `Simulator2D(sensor_params, control_params[, seed])`; opens a new simulation session."""

    def __init__(self, sensor_params, control_params, seed=0, device=0):
        self._ses = _Session._latest = _Session()
        self._ses.sim = self
        self._sensor_params, self._control_params, self._seed, self._device = sensor_params, control_params, int(seed), device
        self._start, self._env_params, self._num_landmarks, self._fixed_landmarks = Pose2(), None, 0, []
        self.sensor_model = BearingRangeSensorModel(sensor_params, seed, self._ses)
        self.control_model = SimpleControlModel(control_params, seed, self._ses)
        self.environment = Environment(self._ses, True)

    def initialize_vehicle(self, pose):
        if self._ses.engine is not None:
            raise RuntimeError("initialize_vehicle after the simulation started")
        self._start = pose

    def random_landmarks(self, landmarks, num, env_params):
        """Simulator2D::addLandmarks (Simulator2D.cpp:445-464): the listed `landmarks` (Point2; the ini file's optional
        [Landmarks] section, pyss2d.py:107-115) take the keys 0 .. k - 1, then `num` landmarks are sampled uniformly in the
        environment box, >= 2 m from the vehicle."""
        self._fixed_landmarks = [(float(p.x), float(p.y)) for p in landmarks]
        self._num_landmarks, self._env_params = len(self._fixed_landmarks) + int(num), env_params
        self.environment.parameter = env_params

    @property
    def vehicle(self):
        p = self._start if self._ses.engine is None else Pose2(*self._ses.engine.ground_truth(0)[0])
        p._session = self._ses  # (carries the simulation into SLAM2D.add_prior)
        return p

    def move(self, odom, ignore_safety=True):
        """Simulator2D::move(odom, ignore_safety) (Simulator2D.cpp:491-503) -> (True = the move was applied, ControlState);
        the reference returns False only when `checkSafety` rejects the new pose, which needs obstacles (none on this
        path: SURVEY.md section 8, `safe_distance` inert).  The SLAM side of the same step
        (`SLAM2D.add_odometry(control_state)`) is part of the same staged call."""
        import torch
        e = self._ses.require_engine()
        e.stage_move(torch.tensor([[odom.x, odom.y, odom.theta]], dtype=torch.float64, device=e.device))
        cp = self._control_params
        cs = SimpleControlModelState(self.vehicle, odom, (cp.translation_noise, cp.translation_noise, cp.rotation_noise))
        cs._session = self._ses
        self._ses.slam._pending_odometry = cs
        return True, cs

    def measure(self):
        """Simulator2D::measure (Simulator2D.cpp:505-527): [(key, Measurement)] of the landmarks that pass the gates."""
        e = self._ses.require_engine()
        keys, br, cnt = e.stage_measure()
        n = int(cnt[0])
        keys, br = keys[0, :n].cpu().numpy(), br[0, :n].cpu().numpy()
        sp = self._sensor_params
        return [(int(k), Measurement(float(b), float(r), (sp.bearing_noise, sp.range_noise))) for k, (b, r) in zip(keys, br)]

    def pprint(self):
        self._sensor_params.pprint()
        self._control_params.pprint()


class SLAM2D(object):
    """src/SS2D.cpp:189-208 over the engine's factor lists and k_slam."""

    def __init__(self, map_params, simulator=None):
        self._ses = _Session.join("slam", self, simulator)
        self._map_params = map_params
        self._pending, self._pending_odometry = [], None
        self.map = Environment(self._ses, False, map_params)

    def add_prior(self, state):
        """SLAM2D::addPrior(VehicleBeliefState) (SLAM2D.cpp:44-57); creates the engine (every parameter is known now)."""
        if self._ses.engine is not None:
            raise RuntimeError("add_prior: the prior is added once, at step 0")
        if getattr(state.pose, "_session", self._ses) is not self._ses:
            raise RuntimeError("add_prior: the prior pose comes from another simulation's Simulator2D.vehicle")
        self._ses.materialise(state)

    def add_odometry(self, control_state):
        """SLAM2D::addOdometry (SLAM2D.cpp:70-89): appended by the staged move that produced `control_state`."""
        if getattr(control_state, "_session", self._ses) is not self._ses:
            raise RuntimeError("add_odometry: the ControlState comes from another simulation's Simulator2D.move")
        if control_state is not self._pending_odometry:
            raise ValueError("add_odometry takes the ControlState of the Simulator2D.move that preceded it")
        self._pending_odometry = None

    def add_measurement(self, key, measurement, *unused):
        self._pending.append((int(key), float(measurement.bearing), float(measurement.range)))

    def optimize(self, update_covariance=True):
        """SLAM2D::optimize (SLAM2D.cpp:374-430): pending measurements are appended, then one iSAM2-policy update with
        all block marginals."""
        import torch
        e = self._ses.require_engine()
        if self._pending_odometry is not None:
            raise RuntimeError("optimize before add_odometry(control_state) of the last move")
        lg = max(e.cfg.num_landmarks, 1)
        if len(self._pending) > lg:
            raise ValueError("more measurements than landmarks in one step")
        keys = torch.zeros(1, lg, dtype=torch.int32)
        br = torch.zeros(1, lg, 2, dtype=torch.float64)
        for k, (key, b, r) in enumerate(self._pending):
            keys[0, k], br[0, k, 0], br[0, k, 1] = key, b, r
        cnt = torch.tensor([len(self._pending)], dtype=torch.int32)
        if self._pending:
            e.stage_add_measurements(keys.to(e.device), br.to(e.device), cnt.to(e.device))
        self._pending = []
        e.stage_optimize()
        e.check_status()
        if self._ses.vm is not None:
            self._ses.vm._fresh = False

    def key_size(self):
        c = self._ses.require_engine().counts(0)
        return c["poses"] + c["landmarks"]

    def adjacency_degree_get(self):
        self._A, self._X = self._ses.require_engine().adjacency(0)

    def adjacency_out(self):
        return self._A

    def features_out(self):
        return self._X.reshape(-1, 1)

    def get_key_points(self, i):
        e = self._ses.require_engine()
        keys, lxy, _ = e.landmarks(0)
        if i < len(keys):
            return [lxy[i][0], lxy[i][1]]
        xyt, _ = e.poses(0)
        return [xyt[i - len(keys)][0], xyt[i - len(keys)][1]]

    def print_graph(self):
        p, k, b, r = self._ses.require_engine().factors(0)
        for row in zip(p, k, b, r):
            print("x%d - l%d: bearing %.6f range %.6f" % row)

    def pprint(self):
        self._map_params.pprint()


class VirtualLandmark(object):
    def __init__(self, point, probability, information, updated):
        self.point, self.probability, self.information, self.updated = point, probability, information, updated

    @property
    def covariance(self):
        return np.linalg.inv(self.information)


class VirtualMap(object):
    """src/SS2D.cpp:217-239 over the engine's virtual-map planes and k_map."""

    def __init__(self, parameter, seed=0, simulator=None):
        self._ses = _Session.join("vm", self, simulator)
        self._params, self._fresh = parameter, False

    def _same_simulation(self, *others):
        for o in others:
            ses = getattr(o, "_ses", None) or getattr(o, "_session", None)
            if ses is not None and ses is not self._ses:
                raise RuntimeError("VirtualMap.update_*: the argument belongs to another simulation")

    def _rebuild(self):
        if not self._fresh:
            self._ses.require_engine().stage_update_map(rebuild=True)
            self._fresh = True

    def update_probability(self, slam, sensor_model):
        """VirtualMap::updateProbability(slam, sensor) (VirtualMap.cpp:61-84).  The device rebuilds occupancy and
        information together (both are functions of the SLAM state only): the rebuild runs at the first of the two
        update calls after an optimise, the second one finds its result in place."""
        self._same_simulation(slam, sensor_model)
        self._rebuild()

    def update_information(self, map_, sensor_model):
        """VirtualMap::updateInformation(map, sensor) (VirtualMap.cpp:256-271)."""
        self._same_simulation(map_, sensor_model)
        self._rebuild()

    def get_parameter(self):
        return self._params

    @property
    def rows(self):
        return self._ses.require_engine().rows

    @property
    def cols(self):
        return self._ses.require_engine().cols

    def get_virtual_landmark_size(self):
        return self.rows * self.cols

    def explored(self):
        return float(self._ses.require_engine().explored()[0])

    def to_array(self):
        return self._ses.require_engine().virtual_map(0)[0]

    def to_cov_trace(self):
        return self._ses.require_engine().virtual_map(0)[2]

    def to_cov_array(self):
        """VirtualMap::toCovArray (VirtualMap.cpp:140-151): (length, angle) grids."""
        ln, an = self._ses.require_engine().cov_array()
        return ln[0].cpu().numpy(), an[0].cpu().numpy()

    def iter_virtual_landmarks(self):
        e = self._ses.require_engine()
        prob, info, _, upd = e.virtual_map(0)
        c = e.cfg
        for v in range(e.rows * e.cols):
            r, col = divmod(v, e.cols)
            pt = Point2((col + 0.5) * c.resolution + c.map_min_x, (r + 0.5) * c.resolution + c.map_min_y)
            yield VirtualLandmark(pt, float(prob.reshape(-1)[v]), info[v], bool(upd[v]))


BearingRangeSensorModelMeasurement = Measurement  # the pybind class name (src/SS2D.cpp:73)
