"""Parameters of the belief step.

Mirrors the ini sections the reference reads in scripts/envs/pyss2d.py:10-55 and
scripts/envs/pyplanner2d.py:24-54 (values: scripts/envs/exploration_env.ini) and the overrides
ExplorationEnv.reset applies (scripts/envs/exploration_env.py:399-407).
"""
import ctypes as C
import functools
import math

import numpy as np


class DrlgxConfig(C.Structure):
    """struct drlgx_config (include/drlgx.h)."""
    _fields_ = [
        ("bearing_noise", C.c_double), ("range_noise", C.c_double), ("min_bearing", C.c_double),
        ("max_bearing", C.c_double), ("min_range", C.c_double), ("max_range", C.c_double),
        ("translation_noise", C.c_double), ("rotation_noise", C.c_double),
        ("env_min_x", C.c_double), ("env_max_x", C.c_double), ("env_min_y", C.c_double), ("env_max_y", C.c_double),
        ("safe_distance", C.c_double),
        ("map_min_x", C.c_double), ("map_max_x", C.c_double), ("map_min_y", C.c_double), ("map_max_y", C.c_double),
        ("resolution", C.c_double), ("sigma0", C.c_double), ("num_samples", C.c_int32),
        ("sigma_x0", C.c_double), ("sigma_y0", C.c_double), ("sigma_theta0", C.c_double), ("num_landmarks", C.c_int32),
        ("angle_weight", C.c_double), ("distance_weight0", C.c_double), ("distance_weight1", C.c_double),
        ("occupancy_threshold", C.c_double), ("max_edge_length", C.c_double), ("algorithm", C.c_int32),
        ("max_poses", C.c_int32), ("max_landmarks", C.c_int32), ("max_factors", C.c_int32), ("max_actions", C.c_int32),
        ("max_snapshots", C.c_int32),
    ]


def _rot2_theta(th):
    """Rot2(th).theta() — the reference's angle setters wrap (Simulation2D.h:52-55,152)."""
    return math.atan2(math.sin(th), math.cos(th))


def default_config(map_size=40, num_landmarks=None, algorithm=0, max_poses=41, max_landmarks=None,
                   max_factors=None, max_actions=None, max_snapshots=1):
    """exploration_env.ini + ExplorationEnv.reset overrides + read_map_params(ext=20).

    Capacities (engine-side, no counterpart in the reference): max_poses per trajectory (any value the LDS tables of
    k_slam_arrow hold, a few hundred), max_landmarks (any count whose tables fit the LDS; the default caps it at 127 because up
    to there the landmark system of the pose-chain solver stays in registers / LDS - pass it explicitly for worlds where an
    episode observes more, e.g. BASELINE config 5), max_actions = the longest line plan the map can produce
    (1-2 rotations + floor(d / max_edge_length) + 1 translations with d <= the diagonal of the vehicle's box)."""
    c = DrlgxConfig()
    c.bearing_noise = _rot2_theta(math.radians(0.5))
    c.range_noise = 0.02
    c.min_bearing = _rot2_theta(math.radians(-179.9))
    c.max_bearing = _rot2_theta(math.radians(179.9))
    c.min_range = 0.1
    c.max_range = 6.0
    c.translation_noise = 0.1
    c.rotation_noise = _rot2_theta(math.radians(0.2))
    h = map_size / 2
    c.env_min_x, c.env_max_x, c.env_min_y, c.env_max_y = -h, h, -h, h
    c.safe_distance = 0.0
    ext = 20.0
    c.map_min_x, c.map_max_x, c.map_min_y, c.map_max_y = -h - ext, h + ext, -h - ext, h + ext
    c.resolution = 2.0
    c.sigma0 = 1.0
    c.num_samples = 1
    c.sigma_x0 = 0.05
    c.sigma_y0 = 0.05
    c.sigma_theta0 = math.radians(0.01)
    c.num_landmarks = int(map_size ** 2 * 0.005) if num_landmarks is None else int(num_landmarks)
    c.angle_weight = 0.4
    c.distance_weight0 = 5.0
    c.distance_weight1 = 2.0
    c.occupancy_threshold = 0.4
    c.max_edge_length = 2.0
    c.algorithm = algorithm
    c.max_poses = max_poses
    c.max_landmarks = max(1, min(c.num_landmarks, 127) if max_landmarks is None else max_landmarks)
    c.max_factors = max_factors if max_factors is not None else max(64, 12 * max_poses)
    if max_actions is None:
        # start poses and frontiers lie in the padded box the start pose is drawn from / the unpadded box respectively
        # (SURVEY.md App. C.2): the diagonal of the larger one bounds every plan
        span = max(map_size, map_size / 2 + 20.0)
        max_actions = int(math.ceil(math.hypot(span, span) / c.max_edge_length)) + 3
    c.max_actions = max_actions
    c.max_snapshots = max_snapshots
    return c


@functools.lru_cache(maxsize=65536)
def start_pose(lo, map_max_x):
    """Start pose of SS2D.__init__ (pyss2d.py:89-95): legacy numpy global-seed stream, PADDED max_x for
    both coordinates.  `np.random.seed(s); np.random.randint(m)` is the first draw of RandomState(s): private
    streams give the reference's values without touching the caller's global one (and without the two 2.5 KB state copies
    that saving / restoring it cost per call: 43 us x one call per re-created env)."""
    m = int(map_max_x)
    x0 = float(np.random.RandomState(lo + 1).randint(m) - map_max_x / 2)
    y0 = float(np.random.RandomState(lo + 2).randint(m) - map_max_x / 2)
    theta0 = math.radians(float(np.random.RandomState(lo + 3).randint(360)))
    return x0, y0, theta0
