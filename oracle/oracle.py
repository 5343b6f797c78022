"""ctypes binding + Python-side restatement for the drlgx CPU ORACLE — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  `OracleEnv` restates `scripts/envs/exploration_env.py` (ExplorationEnv)
on top of the C++ oracle (oracle/drlgx_oracle.cpp); citations are file:line under /root/reference.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libdrlgx_oracle.so")


class OrcConfig(C.Structure):
    _fields_ = [
        ("bearing_noise", C.c_double), ("range_noise", C.c_double), ("min_bearing", C.c_double),
        ("max_bearing", C.c_double), ("min_range", C.c_double), ("max_range", C.c_double),
        ("translation_noise", C.c_double), ("rotation_noise", C.c_double),
        ("env_min_x", C.c_double), ("env_max_x", C.c_double), ("env_min_y", C.c_double), ("env_max_y", C.c_double),
        ("safe_distance", C.c_double),
        ("map_min_x", C.c_double), ("map_max_x", C.c_double), ("map_min_y", C.c_double), ("map_max_y", C.c_double),
        ("resolution", C.c_double), ("sigma0", C.c_double), ("num_samples", C.c_int),
        ("sigma_x0", C.c_double), ("sigma_y0", C.c_double), ("sigma_theta0", C.c_double), ("num_landmarks", C.c_int),
        ("angle_weight", C.c_double), ("distance_weight0", C.c_double), ("distance_weight1", C.c_double),
        ("occupancy_threshold", C.c_double), ("max_edge_length", C.c_double), ("algorithm", C.c_int),
    ]


def build(force=False):
    """Compile the oracle shared library (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "drlgx_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig), C.c_uint32, C.c_double, C.c_double, C.c_double]
        L.orc_create_fixed.restype = C.c_void_p
        L.orc_create_fixed.argtypes = [C.POINTER(OrcConfig), C.c_uint32, C.c_double, C.c_double, C.c_double, dp, C.c_int]
        L.orc_create_prior.restype = C.c_void_p
        L.orc_create_prior.argtypes = [C.POINTER(OrcConfig), C.c_uint32, C.c_double, C.c_double, C.c_double, dp]
        L.orc_create_prior_pose.restype = C.c_void_p
        L.orc_create_prior_pose.argtypes = [C.POINTER(OrcConfig), C.c_uint32, C.c_double, C.c_double, C.c_double, dp, dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_clone.restype = C.c_void_p
        L.orc_clone.argtypes = [C.c_void_p]
        L.orc_simulate.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_utility.restype = C.c_double
        L.orc_utility.argtypes = [C.c_void_p, C.c_double]
        L.orc_uncertainty_em.restype = C.c_double
        L.orc_uncertainty_em.argtypes = [C.c_void_p, C.c_int]
        L.orc_explored.restype = C.c_double
        L.orc_explored.argtypes = [C.c_void_p]
        L.orc_line_plan.argtypes = [C.c_void_p, C.c_double, C.c_double, dp, C.c_int]
        L.orc_simulations_reward.restype = C.c_double
        L.orc_simulations_reward.argtypes = [C.c_void_p, dp, C.c_int]
        for f in ("orc_step_count", "orc_num_poses", "orc_num_landmarks", "orc_num_factors", "orc_vm_rows",
                  "orc_vm_cols"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_get_poses.argtypes = [C.c_void_p, dp, dp]
        L.orc_get_landmarks.argtypes = [C.c_void_p, ip, dp, dp]
        L.orc_get_cov_traces.argtypes = [C.c_void_p, dp, dp]
        L.orc_get_virtual_map.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_uint8)]
        L.orc_get_gt.argtypes = [C.c_void_p, dp, dp, ip]
        L.orc_get_adjacency.argtypes = [C.c_void_p, dp, dp]
        L.orc_get_factors.argtypes = [C.c_void_p, ip, ip, dp, dp]
        L.orc_get_full_cov.argtypes = [C.c_void_p, dp]
        L.orc_get_slot_keys.argtypes = [C.c_void_p, ip]
        L.orc_get_isam.argtypes = [C.c_void_p, dp, dp, dp, dp, ip]
        L.orc_kat_rng.argtypes = [C.c_uint32, C.c_int, C.c_int, dp]
        L.orc_kat_ci.argtypes = [dp, dp, dp]
        L.orc_kat_occupancy_ladder.argtypes = [C.c_int, dp, C.c_int]
        L.orc_kat_predict.argtypes = [dp, dp, C.c_double, C.c_double, C.POINTER(OrcConfig), dp]
        L.orc_wrap_theta.restype = C.c_double
        L.orc_wrap_theta.argtypes = [C.c_double]
        L.orc_landmark_iteration_order.argtypes = [C.c_int, ip]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def wrap_theta(th):
    return math.atan2(math.sin(th), math.cos(th))


def default_config(map_size=40, num_landmarks=None, algorithm=0):
    """scripts/envs/exploration_env.ini values with ExplorationEnv.reset overrides
    (exploration_env.py:399-407) and read_map_params(ext=20) (pyss2d.py:48-55)."""
    c = OrcConfig()
    c.bearing_noise = wrap_theta(math.radians(0.5))
    c.range_noise = 0.02
    c.min_bearing = wrap_theta(math.radians(-179.9))
    c.max_bearing = wrap_theta(math.radians(179.9))
    c.min_range = 0.1
    c.max_range = 6.0
    c.translation_noise = 0.1
    c.rotation_noise = wrap_theta(math.radians(0.2))
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
    c.num_landmarks = int(map_size ** 2 * 0.005) if num_landmarks is None else num_landmarks
    c.angle_weight = 0.4
    c.distance_weight0 = 5.0
    c.distance_weight1 = 2.0
    c.occupancy_threshold = 0.4
    c.max_edge_length = 2.0
    c.algorithm = algorithm
    return c


def start_pose(lo, map_max_x):
    """pyss2d.py:89-95 — legacy numpy global-seed stream; uses the PADDED max_x for x and y."""
    m = int(map_max_x)
    np.random.seed(lo + 1)
    x0 = float(np.random.randint(m) - map_max_x / 2)
    np.random.seed(lo + 2)
    y0 = float(np.random.randint(m) - map_max_x / 2)
    np.random.seed(lo + 3)
    theta0 = math.radians(float(np.random.randint(360)))
    return x0, y0, theta0


class OracleSim(object):
    """EMExplorer / SS2D facade (scripts/envs/pyss2d.py:58-206, pyplanner2d.py:56-81) on the C++ oracle."""

    def __init__(self, cfg, seed, lo, handle=None, start=None, fixed_landmarks=None, prior_information=None, prior_pose=None):
        """fixed_landmarks: [(x, y)] of the ini file's optional [Landmarks] section (pyss2d.py:107-115): keys 0 .. k - 1, the
        random landmarks follow; cfg.num_landmarks is the total."""
        self.cfg = cfg
        self.L = lib()
        if handle is None:
            x0, y0, th0 = start_pose(lo, cfg.map_max_x) if start is None else start
            if prior_pose is not None:  # SLAM2D.add_prior(VehicleBeliefState(pose, information)) at a pose that is not the vehicle's
                info = None if prior_information is None else np.ascontiguousarray(prior_information, dtype=np.float64).reshape(9)
                pp = np.ascontiguousarray(prior_pose, dtype=np.float64).reshape(3)
                dpp = C.POINTER(C.c_double)
                self.h = C.c_void_p(self.L.orc_create_prior_pose(C.byref(cfg), seed, x0, y0, th0,
                                                                 info.ctypes.data_as(dpp) if info is not None else None, pp.ctypes.data_as(dpp)))
            elif prior_information is not None:  # SLAM2D.add_prior(VehicleBeliefState(pose, information)) with a full 3 x 3 matrix
                info = np.ascontiguousarray(prior_information, dtype=np.float64).reshape(9)
                self.h = C.c_void_p(self.L.orc_create_prior(C.byref(cfg), seed, x0, y0, th0, info.ctypes.data_as(C.POINTER(C.c_double))))
            elif fixed_landmarks is not None and len(fixed_landmarks):
                xy = np.ascontiguousarray(fixed_landmarks, dtype=np.float64).reshape(-1, 2)
                self.h = C.c_void_p(self.L.orc_create_fixed(C.byref(cfg), seed, x0, y0, th0, xy.ctypes.data_as(C.POINTER(C.c_double)), len(xy)))
            else:
                self.h = C.c_void_p(self.L.orc_create(C.byref(cfg), seed, x0, y0, th0))
        else:
            self.h = handle

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def clone(self):
        return OracleSim(self.cfg, 0, 0, handle=C.c_void_p(self.L.orc_clone(self.h)))

    # --- life-cycle
    def simulate(self, odom):
        return self.L.orc_simulate(self.h, float(odom[0]), float(odom[1]), float(odom[2]))

    @property
    def step(self):
        return self.L.orc_step_count(self.h)

    # --- planner
    def calculate_utility(self, distance):
        return self.L.orc_utility(self.h, float(distance))

    def uncertainty_em(self, algorithm):
        return self.L.orc_uncertainty_em(self.h, int(algorithm))

    def line_plan(self, goal):
        out = np.zeros((64, 3))
        n = self.L.orc_line_plan(self.h, float(goal[0]), float(goal[1]), _dp(out), 64)
        return out[:n].copy()

    def simulations_reward(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64)
        return self.L.orc_simulations_reward(self.h, _dp(a), len(a))

    # --- getters
    def explored(self):
        return self.L.orc_explored(self.h)

    def num_poses(self):
        return self.L.orc_num_poses(self.h)

    def num_landmarks(self):
        return self.L.orc_num_landmarks(self.h)

    def key_size(self):
        return self.num_poses() + self.num_landmarks()

    def poses(self):
        P = self.num_poses()
        xyt = np.zeros((P, 3))
        info = np.zeros((P, 3, 3))
        self.L.orc_get_poses(self.h, _dp(xyt), _dp(info))
        return xyt, info

    def landmarks(self):
        n = self.num_landmarks()
        keys = np.zeros(n, dtype=np.int32)
        xy = np.zeros((n, 2))
        info = np.zeros((n, 2, 2))
        self.L.orc_get_landmarks(self.h, _ip(keys), _dp(xy), _dp(info))
        return keys, xy, info

    def cov_traces(self):
        lm = np.zeros(self.num_landmarks())
        ps = np.zeros(self.num_poses())
        self.L.orc_get_cov_traces(self.h, _dp(lm), _dp(ps))
        return lm, ps

    def vm_shape(self):
        return self.L.orc_vm_rows(self.h), self.L.orc_vm_cols(self.h)

    def virtual_map(self):
        r, c = self.vm_shape()
        V = r * c
        prob = np.zeros(V)
        info = np.zeros((V, 2, 2))
        tr = np.zeros(V)
        upd = np.zeros(V, dtype=np.uint8)
        self.L.orc_get_virtual_map(self.h, _dp(prob), _dp(info), _dp(tr), upd.ctypes.data_as(C.POINTER(C.c_uint8)))
        return prob.reshape(r, c), info, tr.reshape(r, c), upd

    def ground_truth(self):
        veh = np.zeros(3)
        n = self.cfg.num_landmarks
        lms = np.zeros((n, 2))
        order = np.zeros(n, dtype=np.int32)
        self.L.orc_get_gt(self.h, _dp(veh), _dp(lms), _ip(order))
        return veh, lms, order

    def adjacency(self):
        N = self.key_size()
        A = np.zeros((N, N))
        X = np.zeros(N)
        self.L.orc_get_adjacency(self.h, _dp(A), _dp(X))
        return A, X

    def factors(self):
        m = self.L.orc_num_factors(self.h)
        pose = np.zeros(m, dtype=np.int32)
        key = np.zeros(m, dtype=np.int32)
        b = np.zeros(m)
        r = np.zeros(m)
        self.L.orc_get_factors(self.h, _ip(pose), _ip(key), _dp(b), _dp(r))
        return pose, key, b, r

    def full_covariance(self):
        """(n x n covariance of the last solve, L, P): order [landmarks by slot (2 each), poses (3 each)]."""
        n = self.L.orc_get_full_cov(self.h, None)
        out = np.zeros((n, n))
        self.L.orc_get_full_cov(self.h, _dp(out))
        return out, self.num_landmarks(), self.num_poses()

    def slot_keys(self):
        """Ground-truth key of every landmark slot (order of first sighting = the order of the iSAM state arrays)."""
        out = np.zeros(max(self.num_landmarks(), 1), dtype=np.int32)
        self.L.orc_get_slot_keys(self.h, _ip(out))
        return out[:self.num_landmarks()]

    def isam_state(self):
        P, Ln = self.num_poses(), self.num_landmarks()
        thp = np.zeros((P, 3)); dp = np.zeros((P, 3)); thl = np.zeros((Ln, 2)); dl = np.zeros((Ln, 2))
        cnt = C.c_int(0)
        self.L.orc_get_isam(self.h, _dp(thp), _dp(dp), _dp(thl), _dp(dl), C.byref(cnt))
        return thp, dp, thl, dl, cnt.value

    def knife_edge_cells(self, eps=1e-9):
        """Cells whose occupancy / information membership is decided by a margin below `eps` for some
        pose (range == max_range, bearing == FOV limit, range == min_range): floating-point noise decides
        them in the reference itself (e.g. the four cells at exactly 6 m from an integer start pose when
        the trajectory is pure dead reckoning).  Returns a boolean [rows*cols] mask."""
        xyt, _ = self.poses()
        r, c = self.vm_shape()
        cfg = self.cfg
        cx = cfg.map_min_x + cfg.resolution * (np.arange(c) + 0.5)
        cy = cfg.map_min_y + cfg.resolution * (np.arange(r) + 0.5)
        X, Y = np.meshgrid(cx, cy)
        mask = np.zeros((r, c), dtype=bool)
        for x, y, t in xyt:
            dx, dy = X - x, Y - y
            rng = np.sqrt(dx * dx + dy * dy)
            b = np.arctan2(-math.sin(t) * dx + math.cos(t) * dy, math.cos(t) * dx + math.sin(t) * dy)
            near = rng < cfg.max_range + eps
            mask |= np.abs(rng - cfg.max_range) < eps
            mask |= near & (np.abs(rng - cfg.min_range) < eps)
            mask |= near & ((np.abs(b - cfg.max_bearing) < eps) | (np.abs(b - cfg.min_bearing) < eps))
        return mask.reshape(-1)

    def key_points(self):
        """SLAM2D::get_key_points for every node (landmarks by key, then poses) — SLAM2D.cpp:152-166."""
        _, lxy, _ = self.landmarks()
        pxyt, _ = self.poses()
        return np.concatenate([lxy, pxyt[:, :2]], axis=0)


class OracleEnv(object):
    """scripts/envs/exploration_env.py ExplorationEnv restated (TEST seeding only)."""

    def __init__(self, map_size, env_index, num_landmarks=None, algorithm=0, start=None):
        self.map_size = map_size
        self.env_index = env_index
        self.start = start  # optional explicit (x, y, theta) instead of the reference's integer start pose
        self.num_landmarks = num_landmarks
        self.algorithm = algorithm
        self.dist = 0.0
        self.ext = 20.0
        self._one_nearest_frontier = False
        self.nearest_frontier_point = 0
        self.loop_clo = False
        self._done = False
        self._frontier = []
        self._frontier_index = []
        self._obs = self.reset()
        self._max_steps = 5000

    # exploration_env.py:389-422
    def reset(self):
        self._done = False
        while True:
            seed1 = seed2 = self.env_index
            self.cfg = default_config(self.map_size, self.num_landmarks, self.algorithm)
            self._sim = OracleSim(self.cfg, seed1, seed2, start=self.start)
            for _ in range(4):
                self._sim.simulate((1, 1, math.pi / 2.0))
            if self._sim.num_landmarks() < 1:
                self.env_index = self.env_index + 50
                continue
            self.map_resolution = self.cfg.resolution
            self.leng_i_map, self.leng_j_map = self._sim.vm_shape()
            return self._get_obs()

    def _get_obs(self):
        self._obs = self._sim.virtual_map()[0]
        return self._obs

    def vehicle_position(self):
        xyt, _ = self._sim.poses()
        return xyt[-1]

    # exploration_env.py:98-105
    def step(self, action):
        self._sim.simulate([action[0], action[1], action[2]])
        self.dist = self.dist + math.sqrt(action[0] ** 2 + action[1] ** 2)
        return self._get_obs(), self.done(), {}

    def status(self):
        return self._sim.explored()

    def done(self):
        return self._done or self._sim.step > self._max_steps or self.status() > 0.85

    # exploration_env.py:170-177
    def get_landmark_error(self, sigma0=1.0):
        keys, xy, _ = self._sim.landmarks()
        _, gt, _ = self._sim.ground_truth()
        error = 0.0
        for k, p in zip(keys, xy):
            error += np.sqrt((gt[k][0] - p[0]) ** 2 + (gt[k][1] - p[1]) ** 2)
        n_gt = self.cfg.num_landmarks
        error += sigma0 * (n_gt - len(keys))
        return error / n_gt

    def get_landmark_size(self):
        return self._sim.num_landmarks()

    # exploration_env.py:190-194
    def max_uncertainty_of_trajectory(self):
        _, X = self._sim.adjacency()
        return np.amax(X[self.get_landmark_size():])

    def index2coor(self, i, j):
        x = (j + 0.5) * self.map_resolution + self.cfg.map_min_x
        y = (i + 0.5) * self.map_resolution + self.cfg.map_min_y
        return [x, y]

    def coor2index(self, x, y):
        map_j = int(round((x - self.cfg.map_min_x) / self.map_resolution - 0.5))
        map_i = int(round((y - self.cfg.map_min_y) / self.map_resolution - 0.5))
        return [map_i, map_j]

    @staticmethod
    def points2dist(p1, p2):
        return np.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2)

    @staticmethod
    def diff_theta(point1, point2, root_theta):
        goal_theta = math.atan2(point1[1] - point2[1], point1[0] - point2[0])
        if goal_theta < 0:
            goal_theta = math.pi * 2 + goal_theta
        if root_theta < 0:
            root_theta = math.pi * 2 + root_theta
        diff = goal_theta - root_theta
        if diff < 0:
            diff = math.pi * 2 + diff
        return diff

    def nearest_frontier(self, point, all_frontiers):
        min_dist = float("Inf")
        min_index = None
        for index, fro in enumerate(all_frontiers):
            d = self.points2dist(point, fro)
            if d < min_dist:
                min_dist = d
                min_index = index
        return min_index

    # exploration_env.py:289-348
    def frontier(self):
        veh = self.vehicle_position()
        vehicle_location = [veh[0], veh[1]]
        a = self._obs < 0.45
        free_i, free_j = np.nonzero(a)
        all_frontiers = []
        kp = self._sim.key_points()
        all_landmarks = [list(kp[k]) for k in range(self.get_landmark_size())]
        self._frontier = []
        self._frontier_index = []
        for ptr in range(len(free_i)):
            ci, cj = free_i[ptr], free_j[ptr]
            count = 0
            i0 = ci - 1 if ci - 1 >= 0 else 0
            i1 = ci + 1 if ci + 1 < self.leng_i_map else self.leng_i_map - 1
            j0 = cj - 1 if cj - 1 >= 0 else 0
            j1 = cj + 1 if cj + 1 < self.leng_j_map else self.leng_j_map - 1
            for ni in range(i0, i1 + 1):
                for nj in range(j0, j1 + 1):
                    if 0.49 < self._obs[ni][nj] < 0.51:
                        count += 1
            if count >= 2:
                xy = self.index2coor(ci, cj)
                if self.cfg.map_min_x + self.ext <= xy[0] <= self.cfg.map_max_x - self.ext and \
                        self.cfg.map_min_y + self.ext <= xy[1] <= self.cfg.map_max_y - self.ext:
                    all_frontiers.append(xy)
        self.all_frontiers = all_frontiers
        cur = all_frontiers[self.nearest_frontier(vehicle_location, all_frontiers)]
        self._frontier.append(cur)
        self._frontier_index.append([0])
        if not self._one_nearest_frontier:
            for ip, p in enumerate(all_landmarks):
                cur = all_frontiers[self.nearest_frontier(p, all_frontiers)]
                try:
                    self._frontier_index[self._frontier.index(cur)].append(ip + 1)
                except ValueError:
                    self._frontier.append(cur)
                    self._frontier_index.append([ip + 1])

    # exploration_env.py:196-281
    def graph_matrix(self):
        self.frontier()
        trace_map = self._sim.virtual_map()[2]
        key_size = self._sim.key_size()
        fro_size = len(self._frontier)
        adjacency, feat = self._sim.adjacency()
        features = feat.reshape(-1, 1)
        adjacency = np.pad(adjacency, ((0, fro_size), (0, fro_size)), 'constant')
        features = np.pad(features, ((0, fro_size), (0, 0)), 'constant')
        veh = self.vehicle_position()
        robot_location = [veh[0], veh[1]]
        kp = self._sim.key_points()
        for i in range(fro_size):
            fp = self._frontier[i]
            for j in range(len(self._frontier_index[i])):
                index_node = self._frontier_index[i][j]
                if index_node == 0:
                    self.nearest_frontier_point = i + key_size
                    d = self.points2dist(fp, robot_location)
                    adjacency[key_size - 1][i + key_size] = d
                    adjacency[i + key_size][key_size - 1] = d
                else:
                    d = self.points2dist(fp, kp[index_node - 1])
                    adjacency[index_node - 1][i + key_size] = d
                    adjacency[i + key_size][index_node - 1] = d
        for i in range(fro_size):
            idx = self.coor2index(self._frontier[i][0], self._frontier[i][1])
            features[key_size + i][0] = trace_map[idx[0]][idx[1]]
        f2 = np.zeros(np.shape(features))
        f5 = np.zeros(np.shape(features))
        f3 = np.zeros(np.shape(features))
        f4 = np.zeros(np.shape(features))
        root_theta = veh[2]
        for i in range(key_size):
            f2[i][0] = self.points2dist(kp[i], robot_location)
            f5[i][0] = self.diff_theta(kp[i], robot_location, root_theta)
            idx = self.coor2index(kp[i][0], kp[i][1])
            f3[i][0] = self._obs[idx[0]][idx[1]]
        for i in range(fro_size):
            fp = self._frontier[i]
            f2[key_size + i][0] = self.points2dist(fp, robot_location)
            f5[key_size + i][0] = self.diff_theta(fp, robot_location, root_theta)
            idx = self.coor2index(fp[0], fp[1])
            f3[key_size + i][0] = self._obs[idx[0]][idx[1]]
        for i in range(key_size - 1):
            f4[i][0] = -1
        f4[key_size - 1][0] = 0
        for i in range(fro_size):
            f4[key_size + i][0] = 1
        features = np.concatenate((features, f2, f5, f3, f4), axis=1)
        return adjacency, features, None, fro_size

    # exploration_env.py:134-143
    def actions_all_goals(self):
        key_size = self._sim.key_size()
        fro_size = len(self._frontier)
        all_actions = [[]] * (key_size + fro_size)
        for i, vi in enumerate(self._frontier):
            all_actions[i + key_size] = self._sim.line_plan(vi)
        return all_actions

    # exploration_env.py:145-162
    def rewards_all_goals(self, all_actions, return_raw=False):
        key_size = self._sim.key_size()
        fro_size = len(self._frontier)
        rewards = [np.nan] * (key_size + fro_size)
        for i, _ in enumerate(self._frontier):
            rewards[i + key_size] = self._sim.simulations_reward(all_actions[i + key_size])
        raw = np.array(rewards, dtype=np.float64)
        act_max = np.nanargmax(rewards)
        if self.nearest_frontier_point == act_max:
            self.loop_clo = False
            rewards = np.interp(rewards, (np.nanmin(rewards), np.nanmax(rewards)), (-1.0, 0.0))
        else:
            self.loop_clo = True
            rewards = np.interp(rewards, (np.nanmin(rewards), np.nanmax(rewards)), (-1.0, 1.0))
        rewards[np.isnan(rewards)] = 0
        if return_raw:
            return rewards, raw
        return rewards


def data_process(adjacency, features):
    """DeepQ.data_process (scripts/policy.py:211-232): dense A -> (edge_index [2,E], edge_attr [E], x)."""
    s_a = adjacency
    edge_index = []
    edge_attr = []
    edge_set = set()
    n0, n1 = np.shape(s_a)
    for a_i in range(n0):
        for a_j in range(n1):
            if (a_i, a_j) in edge_set or (a_j, a_i) in edge_set or s_a[a_i][a_j] == 0:
                continue
            edge_index.append([a_i, a_j])
            edge_attr.append(s_a[a_i][a_j])
            if a_i != a_j:
                edge_index.append([a_j, a_i])
                edge_attr.append(s_a[a_j][a_i])
            edge_set.add((a_i, a_j))
            edge_set.add((a_j, a_i))
    ei = np.transpose(np.array(edge_index, dtype=np.int64)).reshape(2, -1)
    return ei, np.array(edge_attr, dtype=np.float32), np.array(features, dtype=np.float32)


def map_entropy(obs, map_size=40):
    """scripts/test.py:61-74."""
    diff = -(0.5 * np.log(0.5)) * {40: 1200, 60: 1600, 80: 2000, 100: 2400}[map_size]
    entro = 0.0
    for i in range(obs.shape[0]):
        for j in range(obs.shape[1]):
            entro = entro + obs[i][j] * np.log(obs[i][j])
    return -entro - diff
