"""`ss2d` — value types and parameter classes of the reference's pybind module (src/SS2D.cpp:18-258).

Only the part of the module surface the DRL scripts touch is provided (SURVEY.md §8b "minimum export set"):
Pose2 / Point2 / Rot2 with gtsam semantics, the four parameter classes with the same property names, and the belief-state
value types returned by the engine-backed facades in `pyss2d.py` / `pyplanner2d.py`.  The per-object mutators of the
reference (`SLAM2D.add_odometry`, `Simulator2D.move`, ...) are fused into one device step (`SS2D.simulate`); the facades
expose them at that granularity.
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

    @property
    def theta(self):
        return math.atan2(self._s, self._c)

    def __mul__(self, o):
        r = Pose2()
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
    """BearingRangeSensorModel::Measurement (bearing, range)."""

    def __init__(self, bearing, range_):
        self.bearing, self.range = bearing, range_

    def transform_from(self, origin):
        q = Point2(self.range * math.cos(self.bearing), self.range * math.sin(self.bearing))
        return Point2(origin._c * q.x - origin._s * q.y + origin.x, origin._s * q.x + origin._c * q.y + origin.y)
