"""CPU tests: the oracle against closed-form known answers derived from the reference source
(SURVEY.md §8c iii) — RNG streams, start poses, occupancy ladder, covariance intersection, utility of
the untouched map, line-planner lists, unordered_map iteration order."""
import math

import numpy as np
import pytest

from oracle import oracle as O


def test_rng_stream_matches_libstdcxx_kat():
    # mt19937(0): uniform_real(0,1) x2 then normal(0,1) x3, drawn by sequential statements with this
    # image's libstdc++ (SURVEY.md §8c lists the same five values in printf-argument order)
    out = np.zeros(5)
    O.lib().orc_kat_rng(0, 2, 3, O._dp(out))
    assert out[0] == 0.59284461651668263
    assert out[1] == 0.84426574425659828
    assert out[2] == 0.070859237682211876
    assert out[3] == 0.073041402053790033
    assert out[4] == -1.4223258418434019


def test_start_poses_legacy_numpy_stream():
    # pyss2d.py:89-95 with padded max_x = 40 (map 40)
    exp = {0: (17, -5, 152), 1: (-5, 4, 122), 2: (4, -15, 355), 3: (-15, 15, 201)}
    for lo, (x, y, deg) in exp.items():
        x0, y0, th = O.start_pose(lo, 40.0)
        assert (x0, y0) == (x, y)
        assert th == math.radians(float(deg))


def test_occupancy_ladder_closed_form():
    # OccupancyMap.h:10-19 / OccupancyMap.cpp:55-62: free ladder 0.5 -> 0.3 -> 0.15517 -> 0.07297 -> 0.05 (sticks)
    p = np.zeros(7)
    O.lib().orc_kat_occupancy_ladder(7, O._dp(p), 0)
    lo_free = math.log(0.3 / 0.7)
    lo_min = math.log(0.05 / 0.95)
    exp = [0.0, lo_free, 2 * lo_free, 3 * lo_free, lo_min, lo_min, lo_min]
    for a, l in zip(p, exp):
        assert a == pytest.approx(math.exp(l) / (1 + math.exp(l)), rel=1e-15)
    assert p[2] == pytest.approx(0.15517, abs=1e-5) and p[3] == pytest.approx(0.07297, abs=1e-5)
    # landmark cell: +ln(7/3) clamps at MAX_LOGODDS = LOGODDS2PROB(0.95) = 0.72115 -> p = 0.67285 (App. C.1)
    q = np.zeros(3)
    O.lib().orc_kat_occupancy_ladder(3, O._dp(q), 1)
    mx = math.exp(0.95) / (1 + math.exp(0.95))
    assert q[0] == pytest.approx(math.exp(mx) / (1 + math.exp(mx)), rel=1e-15)
    assert q[0] == pytest.approx(0.67285, abs=1e-5)
    assert q[1] == q[0]  # occupied cells stay saturated


def test_covariance_intersection_identities():
    L = O.lib()
    out = np.zeros(4)
    m = np.array([2.0, 0.3, 0.3, 1.0])
    L.orc_kat_ci(O._dp(m), O._dp(m), O._dp(out))  # m1 == m2 -> m1 (d = 0 -> w = nan/inf guarded by weights)
    # a = b, c = 2a -> d = 0, w = 0/0 = nan: reference returns nan*m1 + nan*m2 — we only check finite cases
    m1 = np.array([4.0, 0.0, 0.0, 1.0])
    m2 = np.array([1.0, 0.0, 0.0, 4.0])
    L.orc_kat_ci(O._dp(m1), O._dp(m2), O._dp(out))
    a = b = 4.0
    c = a * (1 / 4 + 4.0)
    d = a + b - c
    w = 0.5 * (2 * b - c) / d
    assert 0 <= w <= 1
    np.testing.assert_allclose(out, w * m1 + (1 - w) * m2, rtol=1e-15)
    # dominated case: m1 >> m2 -> w clamps
    m1 = np.array([100.0, 0.0, 0.0, 100.0])
    m2 = np.array([1.0, 0.0, 0.0, 1.0])
    L.orc_kat_ci(O._dp(m1), O._dp(m2), O._dp(out))
    a, b = 1e4, 1.0
    c = a * 0.02
    d = a + b - c
    w = 0.5 * (2 * b - c) / d
    assert w < 0 and d > 0  # -> w = 1
    np.testing.assert_allclose(out, m1, rtol=1e-15)


def test_untouched_map_utility_is_V_2_sigma0_sq():
    # VirtualMap.cpp:333-336 + Planner2D.cpp:343-366: before any update U = V * 2 * sigma0^2
    cfg = O.default_config(40)
    sim = O.OracleSim(cfg, 0, 0)
    rows, cols = sim.vm_shape()
    assert (rows, cols) == (40, 40)
    assert sim.calculate_utility(0.0) == pytest.approx(1600 * 2.0, rel=1e-14)
    # distance weight: no known cells -> w0 = 5
    assert sim.calculate_utility(2.0) == pytest.approx(3200 + 10.0, rel=1e-14)
    assert sim.explored() == 0.0


def test_landmark_iteration_order_libstdcxx():
    # libstdc++ unordered_map<unsigned,...> with keys 0..n-1 inserted in order (Simulator2D.cpp:331-344)
    order = np.zeros(8, dtype=np.int32)
    n = O.lib().orc_landmark_iteration_order(8, O._ip(order))
    assert n == 8 and list(order) == [7, 6, 5, 4, 3, 2, 1, 0]
    order = np.zeros(100, dtype=np.int32)
    O.lib().orc_landmark_iteration_order(100, O._ip(order))
    assert sorted(order) == list(range(100))


def test_line_planner_lists():
    # Planner2D.cpp:937-1041: rotation (shortest turn) then floor(d/2) steps of (2,0,0) + remainder
    env = O.OracleEnv(40, 0)
    veh = env.vehicle_position()
    for goal in ([veh[0] + 5.0, veh[1]], [veh[0] - 3.0, veh[1] + 4.0], [veh[0], veh[1] - 7.5]):
        acts = env._sim.line_plan(goal)
        d = math.hypot(goal[0] - veh[0], goal[1] - veh[1])
        nfull = int(d / 2.0)
        assert len(acts) == 1 + nfull + 1
        assert acts[0][0] == 0 and acts[0][1] == 0 and abs(acts[0][2]) <= math.pi + 1e-12
        for a in acts[1:1 + nfull]:
            assert tuple(a) == (2.0, 0.0, 0.0)
        assert acts[-1][0] == pytest.approx(d - 2.0 * nfull, abs=1e-12)
        # heading after the rotation points at the goal
        want = math.atan2(goal[1] - veh[1], goal[0] - veh[0])
        got = veh[2] + acts[0][2]
        assert math.atan2(math.sin(got - want), math.cos(got - want)) == pytest.approx(0.0, abs=1e-12)


def test_reward_is_utility_difference_and_invariants():
    env = O.OracleEnv(40, 3)
    A, X, _, fro = env.graph_matrix()
    assert np.allclose(A, A.T) and np.all(np.diag(A) == 0)
    ks = A.shape[0] - fro
    L = env.get_landmark_size()
    # node order landmarks -> poses -> frontiers; type feature -1 / 0 / +1
    assert np.all(X[:ks - 1, 4] == -1) and X[ks - 1, 4] == 0 and np.all(X[ks:, 4] == 1)
    assert 0.0 <= env.status() <= 1.0
    _, info = env._sim.poses()
    for m in info:
        assert np.all(np.linalg.eigvalsh(m) > 0)
    acts = env.actions_all_goals()
    r, raw = env.rewards_all_goals(acts, return_raw=True)
    assert np.all(np.isnan(raw[:ks])) and np.all(np.isfinite(raw[ks:]))
    assert np.all(r[:ks] == 0) and r.min() >= -1.0 and r.max() <= 1.0
    # look-ahead does not mutate the live state (Planner2D.cpp:1417-1420 deep copies)
    u0 = env._sim.calculate_utility(0.0)
    env._sim.simulations_reward(acts[ks])
    assert env._sim.calculate_utility(0.0) == u0


def test_fm2_restatement_equals_information_form():
    """oracle/fm2_ref.py (FastMarginals2::update restated: Sigma' = Sigma - Sigma A^T (I + A Sigma A^T)^-1 A Sigma on
    the updated keys' diagonal blocks, cross covariances of new poses by the F chains) against the same update done in
    information form - append the new variables and add J^T J of the same linearised factors, then invert: the two are
    algebraically identical (the reference's own USE_FAST_MARGINAL2 / isam2.update alternatives, Planner2D.cpp:502-511)."""
    import math
    from oracle import fm2_ref as F
    sim = O.OracleSim(O.default_config(40, num_landmarks=60), 5, 5, start=(-3.2, 4.1, 0.4))
    for a in [(1, 1, math.pi / 2)] * 4 + [(2, 0, 0), (2, 0, 0), (0, 0, 0.9), (2, 0, 0), (1.2, 0, 0)]:
        sim.simulate(a)
    actions = [(0.0, 0.0, -0.7), (2.0, 0.0, 0.0), (2.0, 0.0, 0.0), (0.8, 0.0, 0.0)]
    got, n_meas = F.fm2_update(sim, actions)
    assert sum(n_meas) >= 2  # predicted measurements exist, so the update is not the pure propagation
    Sig, L, P = sim.full_covariance()
    thp, _, thl, _, _ = sim.isam_state()
    cfg = sim.cfg
    K = len(actions)
    n0, n = 2 * L + 3 * P, 2 * L + 3 * (P + K)
    Lam = np.zeros((n, n))
    Lam[:n0, :n0] = np.linalg.inv(Sig)
    sig = np.array([cfg.translation_noise, cfg.translation_noise, cfg.rotation_noise])
    sig_br = np.array([cfg.bearing_noise, cfg.range_noise])
    pidx = lambda i: slice(2 * L + 3 * i, 2 * L + 3 * i + 3)  # noqa: E731
    origin, val0 = F._pose(*sim.poses()[0][-1]), F._pose(*thp[-1])
    keys, est_sorted, _ = sim.landmarks()
    slot_of = {int(k): j for j, k in enumerate(sim.slot_keys())}
    for k, a in enumerate(actions):
        odom = F._pose(*a)
        end = F._compose(origin, odom)
        hx, H1b = F._between(val0, end)
        h, _ = F._between(odom, hx)
        Hl = np.array([[h[2], h[3], 0], [-h[3], h[2], 0], [0, 0, 1.0]])
        J = np.zeros((3, n))
        J[:, pidx(P + k - 1)] = (Hl @ H1b) / sig[:, None]
        J[:, pidx(P + k)] = Hl / sig[:, None]
        Lam += J.T @ J
        for key, xy in zip(keys, est_sorted):
            j = slot_of[int(key)]
            dx, dy = xy[0] - end[0], xy[1] - end[1]
            rng = math.hypot(dx, dy)
            b = math.atan2(-end[3] * dx + end[2] * dy, end[2] * dx + end[3] * dy)
            if rng < cfg.max_range and cfg.min_bearing < b < cfg.max_bearing and cfg.min_range < rng:
                Jx, Jl = F._br_jacobians(end, thl[j])
                J = np.zeros((2, n))
                J[:, pidx(P + k)] = Jx / sig_br[:, None]
                J[:, 2 * j:2 * j + 2] = Jl / sig_br[:, None]
                Lam += J.T @ J
        origin, val0 = end, end
    full = np.linalg.inv(Lam)
    for i in range(P + K):
        np.testing.assert_allclose(got[i], full[pidx(i), pidx(i)], rtol=1e-7, atol=1e-12)
    # with no landmark in view the update degenerates to the propagation cov1 = H1 (H0 cov0 H0^T + I) H1^T
    far = O.OracleSim(O.default_config(40, num_landmarks=0), 1, 1, start=(0.0, 0.0, 0.0))
    far.simulate((2, 0, 0))
    g2, nm = F.fm2_update(far, [(2.0, 0.0, 0.0)])
    assert nm == [0] and np.trace(g2[-1]) > np.trace(g2[-2])


def test_listed_landmarks_take_the_first_keys_and_leave_the_random_stream_alone():
    """Simulator2D::addLandmarks(landmarks, num_random, params) (Simulator2D.cpp:445-464; pyss2d.py:107-118): the listed points
    get the keys 0 .. k - 1 unchecked (also closer than 2 m to the vehicle), the sampled ones follow - the same draws, in the
    same order, as without a list."""
    k, n_random = 3, 9
    fixed = [(1.5, -2.25), (-7.0, 3.5), (0.4, 0.3)]  # the last one is < 2 m from the start: kept, the rule is for sampled points
    start = (0.25, -0.5, 0.3)
    with_list = O.OracleSim(O.default_config(40, num_landmarks=k + n_random), 11, 0, start=start, fixed_landmarks=fixed)
    without = O.OracleSim(O.default_config(40, num_landmarks=n_random), 11, 0, start=start)
    _, gt_a, _ = with_list.ground_truth()
    _, gt_b, _ = without.ground_truth()
    np.testing.assert_array_equal(gt_a[:k], np.array(fixed))
    np.testing.assert_array_equal(gt_a[k:], gt_b)
    assert len(gt_a) == k + n_random
