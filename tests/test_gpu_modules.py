"""GPU tests of the reference-shaped module surface: `ss2d.Simulator2D / SLAM2D / VirtualMap / Environment` and
`planner2d.EMPlanner2D(parameter, sensor_model, control_model)` driven call by call in the order of the reference's
scripts/envs/pyss2d.py and scripts/envs/exploration_env.py, against the fused engine path (bit-identical: the same
kernels launched stage by stage) and against the CPU oracle's ExplorationEnv restatement."""
import ctypes as C
import math
from configparser import ConfigParser

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

MAP = 40


def ini(lo, num=8):
    cp = ConfigParser()
    cp.read_dict({
        "Sensor Model": dict(bearing_noise="0.5", range_noise="0.02", min_bearing="-179.9", max_bearing="179.9",
                             min_range="0.1", max_range="6.0"),
        "Control Model": dict(translation_noise="0.1", rotation_noise="0.2"),
        "Environment": dict(min_x="-20", max_x="20", min_y="-20", max_y="20", max_steps="5000", safe_distance="0.0"),
        "Virtual Map": dict(resolution="2.0", sigma0="1.0", num_samples="1"),
        "Simulator": dict(seed=str(lo), lo=str(lo), num=str(num), sigma_x0="0.05", sigma_y0="0.05", sigma_theta0="0.01"),
        "Planner": dict(seed=str(lo), angle_weight="0.4", distance_weight0="5.0", distance_weight1="2.0", d_weight="0.0",
                        max_edge_length="2.0", max_nodes="0.5", occupancy_threshold="0.4", safe_distance="1.0",
                        algorithm="EM_AOPT", reg_out="false"),
    })
    return cp


def test_module_classes_follow_the_reference_call_sequence():
    """pyss2d.SS2D.__init__ / simulate written against our `ss2d` / `planner2d` modules exactly as the reference writes
    them against its pybind modules (constructor signatures, method names, call order), next to the fused facade."""
    from drl_graph_exploration_amd import planner2d, ss2d
    from drl_graph_exploration_amd.pyplanner2d import EMExplorer, config_from_ini
    lo = 3
    start = tuple(np.array(O.start_pose(lo, MAP / 2 + 20)) + np.array([0.2871, -0.3179, 0.0917]))
    cp = ini(lo)
    _, prm = config_from_ini(cp)
    # ---- the reference's SS2D.__init__ (pyss2d.py:102-138)
    sim = ss2d.Simulator2D(prm["sensor"], prm["control"], lo)
    sim.initialize_vehicle(ss2d.Pose2(*start))
    slam = ss2d.SLAM2D(prm["map"])
    virtual_map = ss2d.VirtualMap(prm["virtual_map"], lo)
    sim.random_landmarks([], 8, prm["environment"])
    initial_state = ss2d.VehicleBeliefState(sim.vehicle, np.diag([1.0 / 0.05 ** 2, 1.0 / 0.05 ** 2, 1.0 / math.radians(0.01) ** 2]))
    slam.add_prior(initial_state)
    for key, m in sim.measure():
        slam.add_measurement(key, m)
    slam.optimize(update_covariance=True)
    planner = planner2d.EMPlanner2D(prm["planner"], sim.sensor_model, sim.control_model)  # pyplanner2d.py:61-62
    fused = EMExplorer(ini(lo), start=start)
    ref = O.OracleSim(O.default_config(MAP), lo, lo, start=start)
    assert virtual_map.explored() == 0.0 and np.all(virtual_map.to_array() == 0.5)  # untouched until the first simulate
    assert planner2d.EMPlanner2D.calculate_utility(virtual_map, 0.0, prm["planner"]) == 3200.0
    script = [(1, 1, math.pi / 2)] * 4 + [(2.0, 0.0, 0.0), (0.0, 0.0, 0.8), (2.0, 0.0, 0.0), (1.1, 0.0, 0.0)]
    for odom in script:
        # ---- the reference's SS2D.simulate (pyss2d.py:171-206)
        moved, control_state = sim.move(ss2d.Pose2(*odom), True)
        slam.add_odometry(control_state)
        assert moved is True and control_state.odom.x == odom[0]  # Simulator2D.cpp:491-503: true = applied
        discarded = sim.measure()  # obstacle logic: only consumes sensor noise at safe_distance = 0
        measurements = sim.measure()
        assert [k for k, _ in discarded] == [k for k, _ in measurements]
        for key, m in measurements:
            slam.add_measurement(key, m)
        slam.optimize(update_covariance=True)
        virtual_map.update_probability(slam, sim.sensor_model)
        virtual_map.update_information(slam.map, sim.sensor_model)
        fused.simulate(odom)
        ref.simulate(odom)
    # staged == fused, bit for bit (same kernels, same order of arithmetic)
    e1, e2 = slam._ses.engine, fused.engine
    for a, b in zip(e1.poses(0) + e1.landmarks(0) + e1.virtual_map(0) + e1.factors(0), e2.poses(0) + e2.landmarks(0) + e2.virtual_map(0) + e2.factors(0)):
        np.testing.assert_array_equal(a, b)
    assert virtual_map.explored() == float(e2.explored()[0]) == ref.explored()
    # ... and the oracle
    np.testing.assert_allclose(e1.poses(0)[0], ref.poses()[0], atol=1e-9)
    v = sim.vehicle
    np.testing.assert_allclose([v.x, v.y, v.theta], ref.ground_truth()[0], atol=1e-12)
    assert slam.key_size() == ref.key_size() and slam.map.get_landmark_size() == ref.num_landmarks()
    assert sim.environment.get_landmark_size() == 8 and slam.map.get_trajectory_size() == len(script) + 1
    cur = slam.map.get_current_vehicle()
    np.testing.assert_allclose(np.linalg.inv(cur.information), cur.covariance)
    # Environment.distance (Simulator2D.cpp:244-250, Distance.cpp:5-9) of the estimated trajectory
    est, want = ref.poses()[0], 0.0
    for a, b in zip(est[:-1], est[1:]):
        dx, dy = b[0] - a[0], b[1] - a[1]
        bear = math.atan2(-math.sin(a[2]) * dx + math.cos(a[2]) * dy, math.cos(a[2]) * dx + math.sin(a[2]) * dy)
        want += math.sqrt(dx * dx + dy * dy + (0.5 * bear) ** 2)
    assert slam.map.distance == pytest.approx(want, abs=1e-8) and fused.distance == pytest.approx(want, abs=1e-8)
    # planner calls (pyplanner2d.py:64-81)
    goal = (cur.pose.x + 3.0, cur.pose.y - 2.0)
    plan = planner.line_planner(slam, virtual_map, slam.key_size(), goal[0], goal[1])
    oplan = ref.line_plan(goal)
    np.testing.assert_allclose([[a.x, a.y, a.theta] for a in plan], oplan, atol=1e-9)
    assert planner.simulations_reward(slam, virtual_map, sim, plan) == pytest.approx(ref.simulations_reward(oplan), abs=1e-6)
    assert planner2d.EMPlanner2D.calculate_utility(virtual_map, 1.5, prm["planner"]) == pytest.approx(ref.calculate_utility(1.5), rel=1e-9)
    # a plan to a graph node (n_key < key_size): the goal is that node's estimate (Planner2D.cpp:963-968)
    to_node = planner.line_planner(slam, virtual_map, 0, 0.0, 0.0)
    kx, ky = slam.get_key_points(0)
    np.testing.assert_allclose([[a.x, a.y, a.theta] for a in to_node], ref.line_plan((kx, ky)), atol=1e-9)
    with pytest.raises(NotImplementedError):
        planner.optimize2(slam, virtual_map)
    # VirtualMap.to_cov_array (VirtualMap.cpp:140-151) against numpy's symmetric eigen-decomposition of each cell
    ln, an = virtual_map.to_cov_array()
    cells = list(virtual_map.iter_virtual_landmarks())
    assert ln.shape == an.shape == (virtual_map.rows, virtual_map.cols) and len(cells) == virtual_map.get_virtual_landmark_size()
    for idx in np.flatnonzero(np.array([c.updated for c in cells]))[::7]:
        w, vec = np.linalg.eigh(cells[idx].covariance)
        r, c = divmod(int(idx), virtual_map.cols)
        assert ln[r, c] == pytest.approx(min(math.sqrt(w[1]), 1.0), rel=1e-9)
        d = math.atan2(vec[1, 1], vec[0, 1]) - an[r, c]  # the eigenvector's sign is a convention: equal modulo pi
        assert abs(math.sin(d)) < 1e-6
    np.testing.assert_allclose(virtual_map.to_cov_trace(), ref.virtual_map()[2], rtol=1e-7)


def test_exploration_env_member_accesses_through_the_facade():
    """The member accesses ExplorationEnv makes on its EMExplorer (scripts/envs/exploration_env.py:82-162, :196-348):
    `_virtual_map.to_cov_array / to_array / to_cov_trace / explored`, `_slam.key_size / adjacency_degree_get /
    adjacency_out / features_out / get_key_points / map.*`, `line_plan`, `simulations_reward`, `calculate_utility`,
    `vehicle_position`, `simulate` - through the module classes driven in the reference's order (tests/staged_facade.py),
    with the oracle's ExplorationEnv as the expected values."""
    from staged_facade import StagedEMExplorer
    lo = 6
    start = tuple(np.array(O.start_pose(lo, MAP / 2 + 20)) + np.array([0.3113, 0.2291, -0.0517]))
    ex = StagedEMExplorer(ini(lo), start=start)
    ref = O.OracleEnv(MAP, lo, start=start)
    for _ in range(4):
        ex.simulate((1, 1, math.pi / 2.0))  # ExplorationEnv.reset (:404-405)
    for decision in range(3):
        # _get_obs (:82-88)
        cov_array = ex._virtual_map.to_cov_array()
        obs = ex._virtual_map.to_array()
        assert cov_array[0].shape == obs.shape
        np.testing.assert_array_equal(obs, ref._sim.virtual_map()[0])
        assert ex._virtual_map.explored() == ref.status()
        # graph_matrix (:196-281): adjacency + features + key points, frontier from the occupancy grid
        ex._slam.adjacency_degree_get()
        A, X, _, fro = ref.graph_matrix()
        ks = ex._slam.key_size()
        assert ks == A.shape[0] - fro and ex._slam.map.get_landmark_size() == ref.get_landmark_size()
        np.testing.assert_allclose(ex._slam.adjacency_out(), A[:ks, :ks], atol=1e-7)
        np.testing.assert_allclose(ex._slam.features_out()[:, 0], X[:ks, 0], rtol=1e-5, atol=1e-9)
        pts = np.array([ex._slam.get_key_points(i) for i in range(ks)])
        np.testing.assert_allclose(pts, ref._sim.key_points(), atol=1e-8)
        vp = ex.vehicle_position
        np.testing.assert_allclose([vp.x, vp.y, vp.theta], ref.vehicle_position(), atol=1e-8)
        np.testing.assert_allclose(ex._virtual_map.to_cov_trace(), ref._sim.virtual_map()[2], rtol=1e-7)
        # actions_all_goals / rewards_all_goals (:134-162)
        acts = ref.actions_all_goals()
        raw = []
        for i, goal in enumerate(ref._frontier):
            plan = ex.line_plan(ks, goal)
            np.testing.assert_allclose([[a.x, a.y, a.theta] for a in plan], acts[ks + i], atol=1e-9)
            raw.append(ex.simulations_reward(plan))
        np.testing.assert_allclose(raw, [ref._sim.simulations_reward(acts[ks + i]) for i in range(fro)], atol=1e-6)
        assert ex.calculate_utility(0.7) == pytest.approx(ref._sim.calculate_utility(0.7), rel=1e-9)
        # step (:98-105) along the first plan
        for a in acts[ks + decision % fro]:
            assert ex.simulate([a[0], a[1], a[2]]) is False
            ref.step(a)


def test_two_simulations_driven_alternately_and_constructed_interleaved():
    """The reference's objects are independent values: two simulations may be constructed interleaved (with explicit
    wiring) and driven alternately; implicit wiring that would cross two simulations raises instead of mis-wiring."""
    from drl_graph_exploration_amd import ss2d
    from drl_graph_exploration_amd.pyplanner2d import EMExplorer, config_from_ini
    from staged_facade import StagedEMExplorer
    los = (2, 5)
    starts = [tuple(np.array(O.start_pose(lo, MAP / 2 + 20)) + np.array([0.21, -0.13, 0.05])) for lo in los]
    # (1) two staged facades + two fused facades, stepped alternately
    st = [StagedEMExplorer(ini(lo), start=s) for lo, s in zip(los, starts)]
    fu = [EMExplorer(ini(lo), start=s) for lo, s in zip(los, starts)]
    ref = [O.OracleSim(O.default_config(MAP), lo, lo, start=s) for lo, s in zip(los, starts)]
    script = [(1, 1, math.pi / 2)] * 3 + [(2.0, 0.0, 0.0), (0.0, 0.0, -0.6), (1.5, 0.0, 0.0)]
    for k, odom in enumerate(script):
        order = (0, 1) if k % 2 == 0 else (1, 0)
        for i in order:
            st[i].simulate(odom)
        for i in reversed(order):
            fu[i].simulate(odom)
            ref[i].simulate(odom)
    for i in range(2):
        np.testing.assert_array_equal(st[i].engine.poses(0)[0], fu[i].engine.poses(0)[0])
        np.testing.assert_array_equal(st[i].engine.virtual_map(0)[1], fu[i].engine.virtual_map(0)[1])
        np.testing.assert_allclose(st[i].engine.poses(0)[0], ref[i].poses()[0], atol=1e-9)
    assert not np.array_equal(st[0].engine.poses(0)[0], st[1].engine.poses(0)[0])
    # (2) interleaved construction, wired explicitly
    prm = [config_from_ini(ini(lo))[1] for lo in los]
    sims = [ss2d.Simulator2D(p["sensor"], p["control"], lo) for p, lo in zip(prm, los)]
    slams = [ss2d.SLAM2D(p["map"], simulator=s) for p, s in zip(prm, sims)]
    vms = [ss2d.VirtualMap(p["virtual_map"], lo, simulator=s) for p, lo, s in zip(prm, los, sims)]
    for i in (1, 0):
        sims[i].initialize_vehicle(ss2d.Pose2(*starts[i]))
        sims[i].random_landmarks([], 8, prm[i]["environment"])
    prior = np.diag([1.0 / 0.05 ** 2, 1.0 / 0.05 ** 2, 1.0 / math.radians(0.01) ** 2])
    with pytest.raises(RuntimeError):  # simulation 0's SLAM2D given simulation 1's vehicle
        slams[0].add_prior(ss2d.VehicleBeliefState(sims[1].vehicle, prior))
    for i in (0, 1):
        slams[i].add_prior(ss2d.VehicleBeliefState(sims[i].vehicle, prior))
    for i in (1, 0):
        for key, m in sims[i].measure():
            slams[i].add_measurement(key, m)
        slams[i].optimize(update_covariance=True)
    for odom in script:
        cs = [sims[i].move(ss2d.Pose2(*odom), True)[1] for i in (0, 1)]
        with pytest.raises(RuntimeError):
            slams[0].add_odometry(cs[1])
        for i in (1, 0):
            slams[i].add_odometry(cs[i])
            sims[i].measure()
            for key, m in sims[i].measure():
                slams[i].add_measurement(key, m)
            slams[i].optimize(update_covariance=True)
        with pytest.raises(RuntimeError):
            vms[0].update_probability(slams[1], sims[1].sensor_model)
        for i in (0, 1):
            vms[i].update_probability(slams[i], sims[i].sensor_model)
            vms[i].update_information(slams[i].map, sims[i].sensor_model)
    for i in range(2):
        np.testing.assert_array_equal(slams[i]._ses.engine.poses(0)[0], fu[i].engine.poses(0)[0])
        np.testing.assert_array_equal(vms[i].to_cov_trace(), fu[i].engine.virtual_map(0)[2])
    # (3) implicit wiring never crosses simulations: a second SLAM2D / VirtualMap for the latest simulator raises
    a = ss2d.Simulator2D(prm[0]["sensor"], prm[0]["control"], 1)
    b = ss2d.Simulator2D(prm[1]["sensor"], prm[1]["control"], 2)
    ss2d.SLAM2D(prm[0]["map"])  # joins b (the latest)
    with pytest.raises(RuntimeError):
        ss2d.SLAM2D(prm[1]["map"])
    ss2d.SLAM2D(prm[0]["map"], simulator=a)


def test_listed_landmarks_of_the_ini_file_through_both_facades():
    """The ini file's optional [Landmarks] section (pyss2d.py:107-118 -> Simulator2D.random_landmarks(landmarks, num, params),
    Simulator2D.cpp:445-464): the listed points are the ground-truth landmarks 0 .. k - 1, the sampled ones follow.  The fused
    facade, the module classes driven in the reference's order and the oracle agree."""
    from drl_graph_exploration_amd.pyplanner2d import EMExplorer
    from staged_facade import StagedEMExplorer
    lo = 4
    start = tuple(np.array(O.start_pose(lo, MAP / 2 + 20)) + np.array([0.1371, 0.2179, -0.0417]))
    fixed = [(start[0] + 3.0, start[1] + 1.0), (start[0] - 2.5, start[1] + 2.0), (start[0] + 1.0, start[1] - 4.0), (start[0] + 9.0, start[1] + 9.0)]

    def cfg_with_list():
        cp = ini(lo, num=6)
        cp.read_dict({"Landmarks": dict(x=str([float(p[0]) for p in fixed]), y=str([float(p[1]) for p in fixed]))})
        return cp
    fused, staged = EMExplorer(cfg_with_list(), start=start), StagedEMExplorer(cfg_with_list(), start=start)
    ref = O.OracleSim(O.default_config(MAP, num_landmarks=6 + len(fixed)), lo, lo, start=start, fixed_landmarks=fixed)
    assert fused.engine.cfg.num_landmarks == staged.engine.cfg.num_landmarks == 10
    for odom in [(1, 1, math.pi / 2)] * 4 + [(2.0, 0.0, 0.0), (0.0, 0.0, 0.8), (2.0, 0.0, 0.0)]:
        fused.simulate(odom)
        staged.simulate(odom)
        ref.simulate(odom)
    for e in (fused.engine, staged.engine):
        veh, lms = e.ground_truth(0)
        oveh, olms, _ = ref.ground_truth()
        np.testing.assert_array_equal(lms[:4], np.array(fixed))
        np.testing.assert_array_equal(lms, olms)
        np.testing.assert_allclose(veh, oveh, atol=1e-12)
        p, k, b, r = e.factors(0)
        op, ok, ob, orr = ref.factors()
        np.testing.assert_array_equal(p, op)
        np.testing.assert_array_equal(k, ok)
        np.testing.assert_allclose(b, ob, atol=1e-12)
        np.testing.assert_allclose(e.poses(0)[0], ref.poses()[0], atol=1e-9)
        np.testing.assert_allclose(e.virtual_map(0)[1], ref.virtual_map()[1], rtol=1e-7, atol=1e-9)
    assert fused.engine.counts(0)["landmarks"] >= 3  # the three listed landmarks next to the start are observed
    for a, b in zip(fused.engine.poses(0) + fused.engine.landmarks(0), staged.engine.poses(0) + staged.engine.landmarks(0)):
        np.testing.assert_array_equal(a, b)


def test_full_prior_information_matrix_through_the_module_classes():
    """SLAM2D.add_prior(VehicleBeliefState(pose, information)) with a non-diagonal information matrix (src/SS2D.cpp:191,
    SLAM2D.cpp:44-57): drlgx_stage_set_prior_information_host behind the module classes, against the oracle."""
    from drl_graph_exploration_amd import ss2d
    from drl_graph_exploration_amd.pyplanner2d import config_from_ini
    lo = 2
    start = tuple(np.array(O.start_pose(lo, MAP / 2 + 20)) + np.array([0.3371, -0.1179, 0.0817]))
    _, prm = config_from_ini(ini(lo))
    Lm = np.array([[20.0, 0, 0], [4.0, 18.0, 0], [-2.0, 3.0, 90.0]])
    info = Lm @ Lm.T  # symmetric positive definite, every off-diagonal entry non-zero
    sim = ss2d.Simulator2D(prm["sensor"], prm["control"], lo)
    sim.initialize_vehicle(ss2d.Pose2(*start))
    slam = ss2d.SLAM2D(prm["map"], simulator=sim)
    vm = ss2d.VirtualMap(prm["virtual_map"], lo, simulator=sim)
    sim.random_landmarks([], 8, prm["environment"])
    slam.add_prior(ss2d.VehicleBeliefState(sim.vehicle, info))
    ref = O.OracleSim(O.default_config(MAP), lo, lo, start=start, prior_information=info)
    for key, m in sim.measure():
        slam.add_measurement(key, m)
    slam.optimize(update_covariance=True)
    for odom in [(1, 1, math.pi / 2)] * 4 + [(2.0, 0.0, 0.0), (0.0, 0.0, 0.8)]:
        _, cs = sim.move(ss2d.Pose2(*odom), True)
        slam.add_odometry(cs)
        sim.measure()
        for key, m in sim.measure():
            slam.add_measurement(key, m)
        slam.optimize(update_covariance=True)
        vm.update_probability(slam, sim.sensor_model)
        vm.update_information(slam.map, sim.sensor_model)
        ref.simulate(odom)
    e = slam._ses.engine
    xyt, pinfo = e.poses(0)
    oxyt, opinfo = ref.poses()
    np.testing.assert_allclose(xyt, oxyt, atol=1e-9)
    np.testing.assert_allclose(pinfo, opinfo, rtol=1e-7, atol=1e-6)
    np.testing.assert_allclose(pinfo[0], opinfo[0], rtol=1e-7)
    assert abs(pinfo[0].reshape(3, 3)[0, 1]) > 1.0  # the coupling of the prior is in the first pose's marginal
    np.testing.assert_allclose(e.virtual_map(0)[1], ref.virtual_map()[1], rtol=1e-7, atol=1e-9)
    with pytest.raises(ValueError):
        s2 = ss2d.Simulator2D(prm["sensor"], prm["control"], lo)
        s2.random_landmarks([], 8, prm["environment"])
        ss2d.SLAM2D(prm["map"], simulator=s2).add_prior(ss2d.VehicleBeliefState(s2.vehicle, np.array([[1.0, 2, 0], [0, 1, 0], [0, 0, 1]])))


@pytest.mark.gpu
def test_prior_pose_that_is_not_the_vehicles_through_the_module_classes():
    """SLAM2D.add_prior(VehicleBeliefState(pose, information)) with a pose OTHER than the simulator's initial vehicle pose
    (src/SS2D.cpp:193, SLAM2D.cpp:44-57: the prior factor's pose and the initial estimate of x0; the simulator keeps its vehicle where
    it is): drlgx_stage_set_prior_pose_host behind the module classes, against the oracle.  The belief starts 0.4 m / 0.05 rad off
    the ground truth and the bearing-range measurements are taken from the true pose."""
    from drl_graph_exploration_amd import ss2d
    from drl_graph_exploration_amd.pyplanner2d import config_from_ini
    lo = 3
    start = tuple(np.array(O.start_pose(lo, MAP / 2 + 20)) + np.array([0.2113, 0.0907, -0.0311]))
    prior = (start[0] + 0.31, start[1] - 0.25, start[2] + 0.05)
    _, prm = config_from_ini(ini(lo))
    sim = ss2d.Simulator2D(prm["sensor"], prm["control"], lo)
    sim.initialize_vehicle(ss2d.Pose2(*start))
    slam = ss2d.SLAM2D(prm["map"], simulator=sim)
    vm = ss2d.VirtualMap(prm["virtual_map"], lo, simulator=sim)
    sim.random_landmarks([], 8, prm["environment"])
    ocfg = O.default_config(MAP)
    info = np.diag([1.0 / ocfg.sigma_x0 ** 2, 1.0 / ocfg.sigma_y0 ** 2, 1.0 / ocfg.sigma_theta0 ** 2])
    slam.add_prior(ss2d.VehicleBeliefState(ss2d.Pose2(*prior), info))
    ref = O.OracleSim(ocfg, lo, lo, start=start, prior_pose=prior)
    gt, _ = slam._ses.engine.ground_truth(0)
    np.testing.assert_allclose(gt[:2], start[:2], atol=1e-12)  # the vehicle stayed where the simulator put it
    assert abs(math.remainder(gt[2] - start[2], 2 * math.pi)) < 1e-12
    for key, m in sim.measure():
        slam.add_measurement(key, m)
    slam.optimize(update_covariance=True)
    for odom in [(1, 1, math.pi / 2)] * 4 + [(2.0, 0.0, 0.0), (0.0, 0.0, 0.8)]:
        _, cs = sim.move(ss2d.Pose2(*odom), True)
        slam.add_odometry(cs)
        sim.measure()
        for key, m in sim.measure():
            slam.add_measurement(key, m)
        slam.optimize(update_covariance=True)
        vm.update_probability(slam, sim.sensor_model)
        vm.update_information(slam.map, sim.sensor_model)
        ref.simulate(odom)
    e = slam._ses.engine
    xyt, pinfo = e.poses(0)
    oxyt, opinfo = ref.poses()
    assert np.abs(xyt[0, :2] - np.array(start[:2])).max() > 0.05  # the belief really started away from the ground truth
    np.testing.assert_allclose(xyt, oxyt, atol=1e-9)
    np.testing.assert_allclose(pinfo, opinfo, rtol=1e-7, atol=1e-6)
    k, xy, linfo = e.landmarks(0)
    ok, oxy, olinfo = ref.landmarks()
    np.testing.assert_array_equal(k, ok)
    np.testing.assert_allclose(xy, oxy, atol=1e-9)
    np.testing.assert_allclose(e.virtual_map(0)[1], ref.virtual_map()[1], rtol=1e-7, atol=1e-9)
    # ... and only between the staged reset and the first measurement
    assert e.L.drlgx_stage_set_prior_pose_host(e.h, 0, (C.c_double * 3)(0.0, 0.0, 0.0)) == -1
