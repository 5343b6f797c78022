"""Test-side driver: the reference's SS2D / EMExplorer call sequences (scripts/envs/pyss2d.py:102-206,
scripts/envs/pyplanner2d.py:57-84) written against the package's `ss2d` / `planner2d` MODULE classes - one C-ABI call per
member call of the reference's pybind objects.  It exists to prove that the module classes can be driven in the
reference's own order and give the fused step's results; the product facade (`pyplanner2d.SS2D`) issues one fused step."""
import math

import numpy as np

from drl_graph_exploration_amd import planner2d, ss2d
from drl_graph_exploration_amd.config import start_pose
from drl_graph_exploration_amd.pyplanner2d import config_from_ini, load_config


class StagedEMExplorer(object):
    def __init__(self, config, device=0, max_poses=256, start=None):
        self._config = load_config(config) if isinstance(config, str) else config
        cfg, prm = config_from_ini(self._config, max_poses=max_poses)
        self._environment_params, self._map_params, self._planner_params = prm["environment"], prm["map"], prm["planner"]
        lo = int(self._config.getfloat("Simulator", "lo"))
        seed = self._config.getint("Simulator", "seed")
        x0, y0, theta0 = start_pose(lo, cfg.map_max_x) if start is None else start
        self._sim = ss2d.Simulator2D(prm["sensor"], prm["control"], seed, device=device)
        self._sim._ses.max_poses = max_poses
        self._sim._ses.planner_params = self._planner_params
        self._sim.initialize_vehicle(ss2d.Pose2(x0, y0, theta0))
        self._slam = ss2d.SLAM2D(self._map_params)
        self._virtual_map = ss2d.VirtualMap(prm["virtual_map"], seed)
        self._sim.random_landmarks([ss2d.Point2(x, y) for x, y in prm["fixed_landmarks"]], self._config.getint("Simulator", "num"),
                                   self._environment_params)  # pyss2d.py:107-118
        sx0, sy0 = self._config.getfloat("Simulator", "sigma_x0"), self._config.getfloat("Simulator", "sigma_y0")
        st0 = math.radians(self._config.getfloat("Simulator", "sigma_theta0"))
        self._slam.add_prior(ss2d.VehicleBeliefState(self._sim.vehicle, np.diag([1.0 / sx0 ** 2, 1.0 / sy0 ** 2, 1.0 / st0 ** 2])))
        self._cleared = True
        self.engine = self._sim._ses.engine
        self.measure()
        self.optimize()
        self._planner = planner2d.EMPlanner2D(self._planner_params, self._sim.sensor_model, self._sim.control_model)

    def move(self, odom):
        _, control_state = self._sim.move(ss2d.Pose2(odom[0], odom[1], odom[2]), True)
        self._slam.add_odometry(control_state)

    def measure(self):
        for key, m in self._sim.measure():
            self._slam.add_measurement(key, m)

    def optimize(self):
        self._slam.optimize(update_covariance=True)

    def simulate(self, odom):
        mp = self._map_params
        if not mp.min_x < odom[0] < mp.max_x or not mp.min_y < odom[1] < mp.max_y:
            return True
        self.move(odom)
        obstacle = False
        measurements = self._sim.measure()  # (the obstacle test: inert at the shipped safe_distance = 0, consumes sensor noise)
        landmarks = [key for key, _ in self._slam.map.iter_landmarks()]
        for key, m in measurements:
            if (self._cleared or key not in landmarks) and m.range < self._environment_params.safe_distance:
                obstacle, self._cleared = True, False
                break
        if not obstacle:
            self._cleared = True
        self.measure()
        self.optimize()
        self._virtual_map.update_probability(self._slam, self._sim.sensor_model)
        self._virtual_map.update_information(self._slam.map, self._sim.sensor_model)
        return obstacle

    @property
    def vehicle_position(self):
        return self._slam.map.get_current_vehicle().pose

    def calculate_utility(self, distance):
        return planner2d.EMPlanner2D.calculate_utility(self._virtual_map, distance, self._planner_params)

    def line_plan(self, goal_key, fron):
        return self._planner.line_planner(self._slam, self._virtual_map, goal_key, fron[0], fron[1])

    def simulations_reward(self, actions):
        return self._planner.simulations_reward(self._slam, self._virtual_map, self._sim, actions)
