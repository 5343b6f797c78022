"""GPU parity tests (run with -m gpu on an MI355X): the HIP belief step through the C ABI against the
CPU oracle on identical seeds.  Topology (landmark keys, factor lists, update flags, occupancy ladder
membership) must be exact; floating-point state within the tolerances stated here:

  * ground-truth poses / measurements (RNG path): 1e-12 abs  (device log() vs glibc log(): <= 1 ulp)
  * pose / landmark estimates: 1e-9 abs;  information / covariance blocks: 1e-7 relative
  * virtual-map information: 1e-7 relative;  utility: 1e-9 relative;  look-ahead rewards: 1e-6 abs
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

MAP = 40
SCRIPT = [(1, 1, math.pi / 2)] * 4 + [(0, 0, 0.7), (2, 0, 0), (2, 0, 0), (1.3, 0, 0), (0, 0, -1.1), (2, 0, 0), (2, 0, 0),
                                      (0.4, 0, 0), (0, 0, 2.9), (2, 0, 0), (2, 0, 0), (2, 0, 0), (0, 0, -0.3), (2, 0, 0)]


def make_engine(n_envs, n_roll=0, num_landmarks=None, max_poses=41):
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    cfg = default_config(MAP, num_landmarks=num_landmarks, max_poses=max_poses)
    return Engine(cfg, n_envs, n_roll), cfg


def generic_starts(n):
    """Start poses off the integer lattice: no cell sits exactly on a range / FOV boundary, so the
    topology comparison can be strict (the reference's integer starts are covered separately)."""
    return np.array([O.start_pose(lo, MAP / 2 + 20) for lo in range(n)]) + np.array([0.3183, -0.2718, 0.1234])


def compare_state(eng, inst, sim, step_name, check_vm=True, mask_knife_edge=False):
    c = eng.counts(inst)
    assert c["poses"] == sim.num_poses(), step_name
    assert c["landmarks"] == sim.num_landmarks(), step_name
    # ground truth (RNG path)
    veh, lms = eng.ground_truth(inst)
    oveh, olms, _ = sim.ground_truth()
    np.testing.assert_allclose(veh, oveh, atol=1e-12, err_msg=step_name)
    np.testing.assert_array_equal(lms, olms)
    # factor topology exact, values 1e-12
    p, k, b, r = eng.factors(inst)
    op, ok, ob, orr = sim.factors()
    np.testing.assert_array_equal(p, op)
    np.testing.assert_array_equal(k, ok)
    np.testing.assert_allclose(b, ob, atol=1e-12)
    np.testing.assert_allclose(r, orr, atol=1e-12)
    # estimates
    xyt, info = eng.poses(inst)
    oxyt, oinfo = sim.poses()
    np.testing.assert_allclose(xyt, oxyt, atol=1e-9, err_msg=step_name)
    np.testing.assert_allclose(info, oinfo, rtol=1e-7, atol=1e-6, err_msg=step_name)
    keys, xy, linfo = eng.landmarks(inst)
    okeys, oxy, olinfo = sim.landmarks()
    np.testing.assert_array_equal(keys, okeys)
    np.testing.assert_allclose(xy, oxy, atol=1e-9)
    np.testing.assert_allclose(linfo, olinfo, rtol=1e-7, atol=1e-6)
    lt, pt = eng.cov_traces(inst)
    olt, opt = sim.cov_traces()
    np.testing.assert_allclose(lt, olt, rtol=1e-8)
    np.testing.assert_allclose(pt, opt, rtol=1e-8)
    if check_vm:
        prob, vinfo, tr, upd = eng.virtual_map(inst)
        oprob, ovinfo, otr, oupd = sim.virtual_map()
        if mask_knife_edge:
            keep = ~sim.knife_edge_cells()
            prob, oprob = prob.reshape(-1)[keep], oprob.reshape(-1)[keep]
            vinfo, ovinfo, tr, otr = vinfo[keep], ovinfo[keep], tr.reshape(-1)[keep], otr.reshape(-1)[keep]
            upd, oupd = upd[keep], oupd[keep]
        np.testing.assert_array_equal(upd, oupd)
        # the occupancy ladder has 6 reachable values: membership must be identical
        np.testing.assert_allclose(prob, oprob, rtol=1e-14, err_msg=step_name)
        np.testing.assert_allclose(vinfo, ovinfo, rtol=1e-7, atol=1e-9, err_msg=step_name)
        np.testing.assert_allclose(tr, otr, rtol=1e-7)


def test_reset_and_scripted_steps_match_oracle():
    n = 8
    eng, cfg = make_engine(n)
    ocfg = O.default_config(MAP)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    assert eng.status() == 0
    assert list(eng.landmark_order()) == list(sims[0].ground_truth()[2])
    for i in range(n):
        compare_state(eng, i, sims[i], "reset env %d" % i, check_vm=False)
    u = eng.utility().cpu().numpy()
    assert np.all(u == 3200.0)
    for s, act in enumerate(SCRIPT):
        odom = torch.tensor([act] * n, dtype=torch.float64, device=eng.device)
        eng.step(odom)
        for sim in sims:
            sim.simulate(act)
        assert eng.status() == 0
        for i in range(n):
            compare_state(eng, i, sims[i], "step %d env %d" % (s, i))
        u = eng.utility().cpu().numpy()
        ex = eng.explored().cpu().numpy()
        dist = torch.full((n,), 1.7, dtype=torch.float64, device=eng.device)
        ud = eng.utility(dist).cpu().numpy()
        for i in range(n):
            assert u[i] == pytest.approx(sims[i].calculate_utility(0.0), rel=1e-9)
            assert ud[i] == pytest.approx(sims[i].calculate_utility(1.7), rel=1e-9)
            assert ex[i] == sims[i].explored()
        aopt = eng.uncertainty_em(0).cpu().numpy()
        dopt = eng.uncertainty_em(1).cpu().numpy()
        for i in range(n):
            assert aopt[i] == pytest.approx(sims[i].uncertainty_em(0), rel=1e-9)
            assert dopt[i] == pytest.approx(sims[i].uncertainty_em(1), rel=1e-7)
    eng.close()


def test_config0_single_env_small_map():
    """BASELINE configs[0]: ONE environment, 20 m map (V = 900 virtual cells), 30 landmarks - the reference's own
    CPU-runnable case: scripted steps, utility, explored fraction, line plans and look-ahead rewards against the oracle."""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    cfg = default_config(20, num_landmarks=30, max_poses=41)
    eng = Engine(cfg, 1, 8)
    assert eng.rows * eng.cols == 900
    ocfg = O.default_config(20, num_landmarks=30)
    start = (1.3183, -2.2718, 0.6234)
    sim = O.OracleSim(ocfg, 5, 5, start=start)
    eng.reset(np.array([0]), np.array([5]), starts=np.array([start]))
    assert float(eng.utility().cpu()[0]) == 2.0 * 900
    for s, act in enumerate(SCRIPT[:14]):
        eng.step(torch.tensor([act], dtype=torch.float64, device=eng.device))
        sim.simulate(act)
        assert eng.status() == 0
        compare_state(eng, 0, sim, "config-0 step %d" % s)
        assert float(eng.utility().cpu()[0]) == pytest.approx(sim.calculate_utility(0.0), rel=1e-9)
        assert float(eng.explored().cpu()[0]) == sim.explored()
    xyt, _ = sim.poses()
    goals = [(xyt[-1, 0] + 3.0, xyt[-1, 1] - 2.0), (xyt[-1, 0] - 4.5, xyt[-1, 1] + 1.0)]
    ce = torch.zeros(2, dtype=torch.int32, device=eng.device)
    actions, n_act = eng.line_plan(ce, torch.tensor(goals, dtype=torch.float64, device=eng.device))
    acts_h, n_h = actions.cpu().numpy(), n_act.cpu().numpy()
    rewards = eng.lookahead(ce, actions, n_act).cpu().numpy()
    for c, g in enumerate(goals):
        oa = sim.line_plan(g)
        assert n_h[c] == len(oa)
        np.testing.assert_allclose(acts_h[c, :len(oa)], oa, atol=1e-9)
        assert rewards[c] == pytest.approx(sim.simulations_reward(acts_h[c, :n_h[c]]), abs=1e-6)
    eng.close()


def test_reference_integer_start_poses():
    """The reference's own start poses (pyss2d.py:89-95) are integers: when the early trajectory is pure
    dead reckoning the 4 x (1,1,pi/2) reset loop returns to the start and four cells sit EXACTLY at
    max_range; those knife-edge cells are decided by round-off in the reference itself and are masked,
    everything else must still agree."""
    n = 8
    eng, cfg = make_engine(n)
    ocfg = O.default_config(MAP)
    sims = [O.OracleSim(ocfg, lo, lo) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), los=np.arange(n))
    for i in range(n):
        compare_state(eng, i, sims[i], "reset env %d" % i, check_vm=False)
    n_masked = 0
    for s, act in enumerate(SCRIPT[:10]):
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        u = eng.utility().cpu().numpy()
        for i in range(n):
            sims[i].simulate(act)
            compare_state(eng, i, sims[i], "step %d env %d" % (s, i), mask_knife_edge=True)
            k = int(sims[i].knife_edge_cells().sum())
            n_masked += k
            if k == 0:
                assert u[i] == pytest.approx(sims[i].calculate_utility(0.0), rel=1e-9)
    assert eng.status() == 0
    assert n_masked > 0  # the degenerate case really occurs with the reference's reset procedure
    eng.close()


def test_active_mask_and_out_of_bounds_odometry():
    n = 4
    eng, cfg = make_engine(n)
    ocfg = O.default_config(MAP)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    act = (1, 1, math.pi / 2)
    active = torch.tensor([1, 0, 1, 0], dtype=torch.uint8, device=eng.device)
    odom = torch.tensor([act] * n, dtype=torch.float64, device=eng.device)
    odom[2, 0] = 1000.0  # SS2D.simulate rejects increments outside the map box (pyss2d.py:173-176)
    eng.step(odom, active)
    sims[0].simulate(act)
    assert sims[2].simulate((1000.0, 1, math.pi / 2)) == 1
    for i in range(n):
        compare_state(eng, i, sims[i], "masked env %d" % i, check_vm=(i == 0))
    eng.close()


def test_lookahead_rewards_match_oracle():
    n = 6
    eng, cfg = make_engine(n, n_roll=32)
    ocfg = O.default_config(MAP)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    for act in SCRIPT[:9]:
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
    # candidate goals around each robot -> line plans (device) vs oracle line plans
    cand_env, goals = [], []
    for i in range(n):
        xyt, _ = sims[i].poses()
        for dx, dy in ((4.0, 1.0), (-3.0, 5.0), (0.5, -6.5), (-7.0, -2.0)):
            cand_env.append(i)
            goals.append((xyt[-1, 0] + dx, xyt[-1, 1] + dy))
    ce = torch.tensor(cand_env, dtype=torch.int32, device=eng.device)
    gl = torch.tensor(goals, dtype=torch.float64, device=eng.device)
    actions, n_act = eng.line_plan(ce, gl)
    acts_h, n_h = actions.cpu().numpy(), n_act.cpu().numpy()
    for c, (i, g) in enumerate(zip(cand_env, goals)):
        oa = sims[i].line_plan(g)
        assert n_h[c] == len(oa)
        np.testing.assert_allclose(acts_h[c, :len(oa)], oa, atol=1e-9)
    rewards = eng.lookahead(ce, actions, n_act).cpu().numpy()
    assert eng.status() == 0
    for c, (i, g) in enumerate(zip(cand_env, goals)):
        want = sims[i].simulations_reward(acts_h[c, :n_h[c]])
        assert rewards[c] == pytest.approx(want, abs=1e-6), (c, i)
    # live environments are untouched by the look-ahead
    for i in range(n):
        compare_state(eng, i, sims[i], "after lookahead env %d" % i)
    # and a second look-ahead gives the same answer (RNG state copied, not consumed)
    rewards2 = eng.lookahead(ce, actions, n_act).cpu().numpy()
    np.testing.assert_array_equal(rewards, rewards2)
    eng.close()


def test_lookahead_across_kernel_variants():
    """Look-ahead from 39-pose trajectories with plans of up to 9 actions: the rollouts start on the fused LDS kernel and
    cross to the register-tile SLAM variant at 43 poses (per-launch variant selection, virtual map rebuilt only at a
    rollout's last action, bounded launch count): rewards against the oracle's simulations_reward."""
    n = 4
    eng, cfg = make_engine(n, n_roll=16, max_poses=60)
    ocfg = O.default_config(MAP)
    starts = np.array([[-7.3183, -6.2718, 0.1234], [3.1, 4.7, 2.2], [6.4, -8.1, -1.0], [-2.2, 9.3, 0.4]])
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
    for s in range(38):
        act = loop[s % len(loop)]
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
    assert eng.status() == 0 and eng.counts(0)["poses"] == 39
    cand_env, goals = [], []
    for i in range(n):
        xyt, _ = sims[i].poses()
        for dx, dy in ((9.0, 7.0), (-11.0, 3.0), (1.5, -2.5)):  # 6, 6-7 and 2-3 actions
            cand_env.append(i)
            goals.append((xyt[-1, 0] + dx, xyt[-1, 1] + dy))
    ce = torch.tensor(cand_env, dtype=torch.int32, device=eng.device)
    gl = torch.tensor(goals, dtype=torch.float64, device=eng.device)
    actions, n_act = eng.line_plan(ce, gl)
    acts_h, n_h = actions.cpu().numpy(), n_act.cpu().numpy()
    assert n_h.max() >= 6  # 39 + 6 > 42
    rewards = eng.lookahead(ce, actions, n_act, max_n_actions=int(n_h.max())).cpu().numpy()
    assert eng.status() == 0
    for c, i in enumerate(cand_env):
        want = sims[i].simulations_reward(acts_h[c, :n_h[c]])
        assert rewards[c] == pytest.approx(want, abs=1e-6), (c, i, n_h[c])
    np.testing.assert_array_equal(rewards, eng.lookahead(ce, actions, n_act).cpu().numpy())  # unbounded entry point
    for i in range(n):
        compare_state(eng, i, sims[i], "after lookahead env %d" % i, mask_knife_edge=True)
    eng.close()


def test_lookahead_soak_random_states_and_goals():
    """Look-ahead rewards for random goals from 12 randomly driven environments (25 steps each, different graph sizes):
    line plans exact to 1e-9, rewards to 1e-6 against the oracle; more candidates (48) than rollout instances (20)."""
    n = 12
    eng, cfg = make_engine(n, n_roll=20, max_poses=60)
    ocfg = O.default_config(MAP)
    rng = np.random.RandomState(77)
    starts = np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(-3.1, 3.1, n)], 1)
    sims = [O.OracleSim(ocfg, 300 + i, 300 + i, start=tuple(starts[i])) for i in range(n)]
    eng.reset(np.arange(n), 300 + np.arange(n), starts=starts)
    menu = [(2.0, 0.0, 0.0), (1.3, 0.0, 0.0), (0.0, 0.0, 0.9), (0.0, 0.0, -1.4), (0.6, 0.0, 0.3), (1.0, 1.0, math.pi / 2)]
    for s in range(25):
        pick = rng.randint(0, len(menu), n)
        pick[rng.rand(n) < 0.5] = 0
        acts = np.array([menu[k] for k in pick])
        eng.step(torch.tensor(acts, dtype=torch.float64, device=eng.device))
        for i, sim in enumerate(sims):
            sim.simulate(tuple(acts[i]))
    assert eng.status() == 0
    cand_env, goals = [], []
    for i in range(n):
        xyt, _ = sims[i].poses()
        for _ in range(4):
            r, th = rng.uniform(1.0, 13.0), rng.uniform(-math.pi, math.pi)
            cand_env.append(i)
            goals.append((xyt[-1, 0] + r * math.cos(th), xyt[-1, 1] + r * math.sin(th)))
    ce = torch.tensor(cand_env, dtype=torch.int32, device=eng.device)
    gl = torch.tensor(goals, dtype=torch.float64, device=eng.device)
    actions, n_act = eng.line_plan(ce, gl)
    acts_h, n_h = actions.cpu().numpy(), n_act.cpu().numpy()
    rewards = eng.lookahead(ce, actions, n_act, max_n_actions=int(n_h.max())).cpu().numpy()
    assert eng.status() == 0
    for c, (i, g) in enumerate(zip(cand_env, goals)):
        oa = sims[i].line_plan(g)
        assert n_h[c] == len(oa)
        np.testing.assert_allclose(acts_h[c, :len(oa)], oa, atol=1e-9)
        assert rewards[c] == pytest.approx(sims[i].simulations_reward(acts_h[c, :n_h[c]]), abs=1e-6), (c, i, n_h[c])
    eng.close()


def test_snapshot_restore_roundtrip():
    n = 3
    eng, cfg = make_engine(n)
    eng.reset(np.arange(n), np.arange(n), los=np.arange(n))
    for act in SCRIPT[:6]:
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
    eng.snapshot(0)
    before = [eng.poses(i)[0].copy() for i in range(n)]
    u0 = eng.utility().cpu().numpy().copy()
    nxt = torch.tensor([SCRIPT[6]] * n, dtype=torch.float64, device=eng.device)
    eng.step(nxt)
    a1 = [eng.poses(i)[0].copy() for i in range(n)]
    eng.restore(0)
    for i in range(n):
        np.testing.assert_array_equal(eng.poses(i)[0], before[i])
    np.testing.assert_array_equal(eng.utility().cpu().numpy(), u0)
    eng.step(nxt)  # same RNG state -> identical continuation
    for i in range(n):
        np.testing.assert_array_equal(eng.poses(i)[0], a1[i])
    eng.close()


def test_many_landmarks_64_node_graphs():
    """BASELINE config 2 shape: 100 landmarks on the 40 m map, graphs grown to ~64 nodes."""
    n = 4
    eng, cfg = make_engine(n, num_landmarks=100)
    ocfg = O.default_config(MAP, num_landmarks=100)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    script = SCRIPT + [(2, 0, 0), (0, 0, 1.2), (2, 0, 0), (2, 0, 0), (0, 0, -2.0), (2, 0, 0), (2, 0, 0), (1, 0, 0)]
    for s, act in enumerate(script):
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
        if s % 5 == 4 or s == len(script) - 1:
            for i in range(n):
                compare_state(eng, i, sims[i], "step %d env %d" % (s, i))
    assert eng.status() == 0
    c = eng.counts(0)
    assert c["poses"] + c["landmarks"] >= 32 and c["factors"] >= 60
    eng.close()


def test_fused_step_kernel_equals_stage_kernels():
    """drlgx_step launches ONE fused kernel (simulate + SLAM + map); with drlgx_timing_enable(2) the same step runs as
    its three stage kernels.  Same code, same order: the two engines must agree bit for bit, step after step."""
    n = 5
    fused, cfg = make_engine(n, num_landmarks=60)
    staged, _ = make_engine(n, num_landmarks=60)
    staged.timing_enable(2)
    starts = generic_starts(n)
    for e in (fused, staged):
        e.reset(np.arange(n), np.arange(n), starts=starts)
    # ... up to the capacity of the fused kernel's dense solver (41 poses: every tile-row count of the sweep, with and
    # without an idle tile row for the wave that inverts the diagonal tiles - the first one under the Schur phase)
    loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
    script = SCRIPT + [loop[k % len(loop)] for k in range(cfg.max_poses - 1 - len(SCRIPT))]
    for s, act in enumerate(script):
        odom = torch.tensor([act] * n, dtype=torch.float64, device=fused.device)
        fused.step(odom)
        staged.step(odom)
        for i in range(n):
            assert fused.counts(i) == staged.counts(i)
            np.testing.assert_array_equal(fused.poses(i)[0], staged.poses(i)[0])
            np.testing.assert_array_equal(fused.poses(i)[1], staged.poses(i)[1])
            for a, b in zip(fused.landmarks(i), staged.landmarks(i)):
                np.testing.assert_array_equal(a, b)
            for a, b in zip(fused.virtual_map(i), staged.virtual_map(i)):
                np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(fused.utility().cpu().numpy(), staged.utility().cpu().numpy())
    assert fused.status() == 0 and staged.status() == 0
    tm = staged.timing_read()
    assert tm["slam"][1] == len(script) and tm["step"][1] == 0  # the staged engine really ran the stage kernels
    assert max(fused.counts(i)["poses"] for i in range(n)) == cfg.max_poses
    fused.close()
    staged.close()


@pytest.mark.parametrize("max_poses", [43, 86, 127, 200, 256])
def test_larger_capacities_use_the_pose_chain_solver(max_poses, monkeypatch):
    """Beyond 42 poses k_slam solves in the pose-chain-first order (csrc/k_slam_arrow.hip: block LDL^T of the odometry
    chain, dense system on the landmarks only): same results.  The engine picks the kernel per launch from its bound on
    the pose counts; DRLGX_VARIANT_BY_CAPACITY=1 makes it pick by `max_poses`, so that a short trajectory exercises it."""
    monkeypatch.setenv("DRLGX_VARIANT_BY_CAPACITY", "1")
    n = 3
    eng, cfg = make_engine(n, max_poses=max_poses)
    ocfg = O.default_config(MAP)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    for i in range(n):
        compare_state(eng, i, sims[i], "chain solver, reset env %d" % i, check_vm=False)
    for s, act in enumerate(SCRIPT[:12]):
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
        if s in (0, 5):
            for i in range(n):
                compare_state(eng, i, sims[i], "chain solver env %d step %d" % (i, s))
    assert eng.status() == 0
    for i in range(n):
        compare_state(eng, i, sims[i], "chain solver env %d" % i)
    eng.close()


def test_capacities_beyond_the_kernels_are_refused():
    """Per-landmark / per-pose tables beyond the LDS: drlgx_create says so instead of overrunning.  (Hundreds of landmarks
    are fine: test_more_than_127_landmarks_per_instance.)"""
    from drl_graph_exploration_amd import _lib, default_config
    from drl_graph_exploration_amd.engine import Engine
    for kw in (dict(num_landmarks=4000, max_landmarks=4000), dict(max_poses=2000)):
        with pytest.raises(_lib.DrlgxError):
            Engine(default_config(MAP, **kw), 2, 0)


def test_variant_follows_the_trajectory_length():
    """One engine with a 90-pose capacity over an 88-pose trajectory: the steps run on the fused fast kernel up to 42
    poses, then on the pose-chain solver - selected per launch from the host's pose-count bound (exact after every
    status check, +1 per step in between) - and agree with the oracle throughout; timing spans tell which kernel ran."""
    n = 2
    eng, cfg = make_engine(n, max_poses=90)
    ocfg = O.default_config(MAP)
    starts = np.array([[-7.3183, -6.2718, 0.1234], [3.1, 4.7, 2.2]])
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
    eng.timing_enable(True)
    fused_steps = {}
    for s in range(87):
        act = loop[s % len(loop)]
        eng.timing_read()
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
        if s % 10 == 9 or s in (40, 41, 42, 56, 57, 83, 84, 86):
            tm = eng.timing_read()
            fused_steps[s] = tm["step"][1] == 1
            assert eng.status() == 0  # also refreshes the bound
            for i in range(n):
                assert eng.counts(i)["poses"] == s + 2
                compare_state(eng, i, sims[i], "env %d step %d" % (i, s), mask_knife_edge=True)
    # poses after step s = s + 2: the fused LDS kernel serves up to 42 poses
    assert fused_steps[39] and fused_steps[40]
    # beyond that the pose-chain solver takes over: also as ONE kernel with the simulator and the map (k_step_arrow), so the
    # spans alone do not tell the two apart - the stage-kernel form of the same steps is exercised by timing mode 2 and by
    # test_larger_capacities_use_the_pose_chain_solver
    assert fused_steps[41] and fused_steps[86]
    eng.close()


@pytest.mark.parametrize("n,steps,cap,checks,num_lm", [(32, 46, 60, (17, 33, 45), None), (6, 104, 110, (50, 70, 90, 103), None),
                                                       (4, 215, 256, (99, 150, 199, 214), None),
                                                       (3, 180, 200, (60, 120, 179), 100)])
def test_random_walk_soak(n, steps, cap, checks, num_lm):
    """Environments driven by independent random action sequences (turns, full and partial forward moves, a few
    out-of-bounds requests) against one oracle instance each, full state comparison at several steps: 32 envs x 46 steps
    (every tile count of the fast path, into the pose-chain solver), 6 envs x 104 steps, 4 envs x 215 steps (the length of
    the reference's own evaluation episodes: 96-197 actions) and 3 envs x 180 steps among 100 landmarks (the landmark
    system of the pose-chain solver grows past 63 landmarks: register-tile sweep from the workspace)."""
    eng, cfg = make_engine(n, max_poses=cap, num_landmarks=num_lm)
    ocfg = O.default_config(MAP, num_landmarks=num_lm)
    rng = np.random.RandomState(20260927)
    starts = np.stack([rng.uniform(-14, 14, n), rng.uniform(-14, 14, n), rng.uniform(-3.1, 3.1, n)], 1)
    sims = [O.OracleSim(ocfg, 100 + i, 100 + i, start=tuple(starts[i])) for i in range(n)]
    eng.reset(np.arange(n), 100 + np.arange(n), starts=starts)
    menu = [(2.0, 0.0, 0.0), (1.3, 0.0, 0.0), (0.0, 0.0, 0.9), (0.0, 0.0, -1.4), (0.6, 0.0, 0.3), (2.0, 0.0, 0.0), (1.0, 1.0, math.pi / 2),
            (75.0, 0.0, 0.0)]  # the last one is rejected by the bounds check of SS2D.simulate
    for s in range(steps):
        pick = rng.randint(0, len(menu), n)
        pick[rng.rand(n) < 0.5] = 0  # mostly forward moves, so that the trajectories spread out
        acts = np.array([menu[k] for k in pick])
        eng.step(torch.tensor(acts, dtype=torch.float64, device=eng.device))
        for i, sim in enumerate(sims):
            sim.simulate(tuple(acts[i]))
        if s in checks:
            assert eng.status() == 0
            for i in range(n):
                compare_state(eng, i, sims[i], "soak env %d step %d" % (i, s), mask_knife_edge=True)
    assert max(eng.counts(i)["poses"] for i in range(n)) >= {60: 40, 110: 88, 256: 180, 200: 150}[cap]
    if num_lm == 100:
        assert max(eng.counts(i)["landmarks"] for i in range(n)) >= 64
    eng.close()


def test_config5_scale_120_pose_graphs():
    """BASELINE config 5 scale: 50 m map, 500 landmarks, graphs grown to ~115 poses / ~100 landmarks (pose-chain solver
    with a ~200 x 200 landmark system swept from the HBM/L2 workspace, k_map works in pose chunks)."""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    n, msize = 2, 50
    cfg = default_config(msize, num_landmarks=500, max_poses=120, max_landmarks=127, max_factors=3600)
    eng = Engine(cfg, n, 0)
    ocfg = O.default_config(msize, num_landmarks=500)
    starts = np.array([[-7.3183, -6.2718, 0.1234], [3.1, 4.7, 2.2]])
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4)]
    for s in range(114):
        act = loop[s % len(loop)]
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
        if s in (87, 113):
            assert eng.status() == 0
            for i in range(n):
                assert eng.counts(i)["poses"] == s + 2
                compare_state(eng, i, sims[i], "config-5 env %d step %d" % (i, s), mask_knife_edge=True)
    eng.close()


def test_more_than_127_landmarks_per_instance():
    """The reference has no landmark cap (SLAM2D.cpp:103-124 grows its map; exploration_env.py:399 sets the count).
    BASELINE config 5's world (50 m, 500 landmarks) observed along a lawn-mower sweep: > 200 landmarks per instance, i.e.
    a landmark system beyond the register-tile sweep of k_slam_arrow (streamed from the workspace), and before that - up to
    42 poses - the dense pose solve with hundreds of landmarks eliminated analytically."""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    n, msize = 2, 50
    cfg = default_config(msize, num_landmarks=500, max_poses=80, max_landmarks=500, max_factors=3600)
    eng = Engine(cfg, n, 0)
    ocfg = O.default_config(msize, num_landmarks=500)
    starts = np.array([[-21.3183, -19.2718, 0.1234], [20.1, 17.7, 3.3]])
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    lane = [(2, 0, 0)] * 16 + [(0.5, 0, math.pi / 2)] + [(2, 0, 0)] * 4 + [(0.5, 0, math.pi / 2)]
    script = (lane + [(2, 0, 0)] * 16 + [(0.5, 0, -math.pi / 2)] + [(2, 0, 0)] * 4 + [(0.5, 0, -math.pi / 2)] + lane)[:66]
    for s, act in enumerate(script):
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
        if s in (30, 47, 65):  # 32 poses: dense pose solve; 49 / 67 poses: pose-chain solver
            assert eng.status() == 0
            for i in range(n):
                assert eng.counts(i)["poses"] == s + 2
                compare_state(eng, i, sims[i], "many landmarks env %d step %d" % (i, s), mask_knife_edge=True)
    seen = [eng.counts(i)["landmarks"] for i in range(n)]
    assert min(seen) > 200, seen
    eng.close()


@pytest.mark.parametrize("fov_deg,max_range,num_lm", [(60.0, 6.0, 40), (110.0, 4.5, 40)])
def test_narrow_field_of_view_uses_exact_paths(fov_deg, max_range, num_lm):
    """With a narrow sensor the sector-sweep bounding box and the atan2 field-of-view test are no longer
    provably redundant: k_map must take the exact paths (and a different cell window) and still agree."""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    n = 4
    cfg = default_config(MAP, num_landmarks=num_lm)
    ocfg = O.default_config(MAP, num_landmarks=num_lm)
    for c in (cfg, ocfg):
        c.min_bearing = -math.radians(fov_deg)
        c.max_bearing = math.radians(fov_deg)
        c.max_range = max_range
    eng = Engine(cfg, n, 0)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    for s, act in enumerate(SCRIPT):
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for sim in sims:
            sim.simulate(act)
        if s % 4 == 3:
            for i in range(n):
                compare_state(eng, i, sims[i], "fov %g step %d env %d" % (fov_deg, s, i))
    assert eng.status() == 0
    eng.close()


def test_fm2_covariance_update_matches_the_restatement():
    """drlgx_fm2_update (FastMarginals2::update / propagate fed as updateNodeInformation_EM / updateTrajectory_EM feed it)
    against oracle/fm2_ref.py on the same beliefs and action lists: every pose's updated 3x3 covariance, rel 1e-6
    (the prior joint covariance is a dense inverse on both sides)."""
    from oracle import fm2_ref
    n = 4
    eng, cfg = make_engine(n, num_landmarks=60, max_poses=64)
    ocfg = O.default_config(MAP, num_landmarks=60)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    eng.reset(np.arange(n), np.arange(n), starts=starts)
    for act in SCRIPT[:11]:
        eng.step(torch.tensor([act] * n, dtype=torch.float64, device=eng.device))
        for s in sims:
            s.simulate(act)
    assert eng.status() == 0
    plans = [[(0.0, 0.0, -0.7), (2.0, 0.0, 0.0), (2.0, 0.0, 0.0), (0.8, 0.0, 0.0)], [(0.0, 0.0, 2.4), (2.0, 0.0, 0.0)],
             [(0.0, 0.0, 0.3), (2.0, 0.0, 0.0), (2.0, 0.0, 0.0), (2.0, 0.0, 0.0), (2.0, 0.0, 0.0), (1.1, 0.0, 0.0)]]
    cand_env, acts, nact = [], [], []
    for e in range(n):
        for p in plans:
            cand_env.append(e)
            a = np.zeros((cfg.max_actions, 3))
            a[:len(p)] = p
            acts.append(a)
            nact.append(len(p))
    ce = torch.tensor(cand_env, dtype=torch.int32, device=eng.device)
    cov, n_out = eng.fm2_update(ce, torch.tensor(np.array(acts), device=eng.device), torch.tensor(nact, dtype=torch.int32, device=eng.device))
    assert eng.status() == 0
    cov, n_out = cov.cpu().numpy(), n_out.cpu().numpy()
    seen_meas = 0
    for c, (e, k) in enumerate(zip(cand_env, nact)):
        want, n_meas = fm2_ref.fm2_update(sims[e], plans[c % len(plans)])
        seen_meas += sum(n_meas)
        assert n_out[c] == want.shape[0] == sims[e].num_poses() + k
        np.testing.assert_allclose(cov[c, :n_out[c]], want, rtol=1e-6, atol=1e-10, err_msg="candidate %d" % c)
        # the update only ever removes uncertainty from the old poses, and the belief itself is untouched
        prior = np.linalg.inv(sims[e].poses()[1])
        assert all(np.trace(cov[c, i]) <= np.trace(prior[i]) * (1 + 1e-9) for i in range(sims[e].num_poses()))
    assert seen_meas > 20
    for i in range(n):
        compare_state(eng, i, sims[i], "after fm2 env %d" % i)
    eng.close()


def test_incremental_update_equals_the_full_solve_across_relinearisations(monkeypatch):
    """The rank-k covariance update between relinearisations (csrc/k_inc.hip; FastMarginals.cpp:188-321 applied to
    SLAM2D::optimize) against (a) the same engine with every update a full solve (DRLGX_INCREMENTAL=0) and (b) the CPU
    oracle, over 31 consecutive steps - three 10th-update relinearisation checks (SLAM2D.cpp:10-12) - in the 100-landmark
    world: estimates 1e-9, information blocks 1e-7 relative, factor topology and the virtual map's update flags exact.
    The virtual-map information is compared at 1e-6 here: at this many poses the FULL solve itself is 1.5-2.1e-7 away from
    the oracle in a few cells next to a pose (scripts/inc_drift.py prints both engines' errors side by side - the
    incremental path is not the less accurate one)."""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    n = 6
    cfg = default_config(MAP, num_landmarks=100, max_poses=41, max_landmarks=100)
    eng = Engine(cfg, n, 0)
    monkeypatch.setenv("DRLGX_INCREMENTAL", "0")
    ref = Engine(cfg, n, 0)
    monkeypatch.delenv("DRLGX_INCREMENTAL")
    assert ref.inc_stats() == (-1, -1) and eng.inc_stats()[0] == 0
    ocfg = O.default_config(MAP, num_landmarks=100)
    starts = generic_starts(n)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    for e in (eng, ref):
        e.reset(np.arange(n), np.arange(n), starts=starts)
    script = [(1, 1, math.pi / 2)] * 4 + [(2, 0, 0), (2, 0, 0), (0, 0, 0.6)] * 9
    for s, act in enumerate(script):
        odom = torch.tensor([act] * n, dtype=torch.float64, device=eng.device)
        eng.step(odom)
        ref.step(odom)
        for sim in sims:
            sim.simulate(act)
        assert eng.status() == 0 and ref.status() == 0
        for i in range(n):
            name = "incremental, step %d env %d" % (s, i)
            compare_state(eng, i, sims[i], name, check_vm=False)
            xyt, info = eng.poses(i)
            rxyt, rinfo = ref.poses(i)
            np.testing.assert_allclose(xyt, rxyt, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(info, rinfo, rtol=1e-7, atol=1e-6, err_msg=name)
            k, xy, linfo = eng.landmarks(i)
            rk, rxy, rlinfo = ref.landmarks(i)
            np.testing.assert_array_equal(k, rk)
            np.testing.assert_allclose(xy, rxy, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(linfo, rlinfo, rtol=1e-7, atol=1e-6, err_msg=name)
            prob, vinfo, tr, upd = eng.virtual_map(i)
            oprob, ovinfo, otr, oupd = sims[i].virtual_map()
            np.testing.assert_array_equal(upd, oupd)
            np.testing.assert_array_equal(upd, ref.virtual_map(i)[3])
            np.testing.assert_allclose(prob, oprob, rtol=1e-14, err_msg=name)
            np.testing.assert_allclose(vinfo, ovinfo, rtol=1e-6, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(tr, otr, rtol=1e-6, err_msg=name)
    inc, full = eng.inc_stats()
    # every env: the reset solve and the updates that relinearise are full solves, all the others rank-k updates
    assert inc + full == n * (len(script) + 1)
    assert inc >= n * (len(script) - 4) and full >= n
    eng.close()
    ref.close()


def test_compact_map_kernel_equals_the_resident_one():
    """The stand-alone map kernel has a second form for launches with more instances than CUs (csrc/k_map.hip, k_map_c: <= 128
    VGPRs, <= 80 KB of LDS - compact stage, one mask per cell - so that two workgroups share a CU).  Same arithmetic in the same
    order: virtual map, traces and the utility sums must be bit-equal to the one-workgroup-per-CU kernel, step after step."""
    n = 5
    a, cfg = make_engine(n, num_landmarks=60)
    b, _ = make_engine(n, num_landmarks=60)
    for e in (a, b):
        e.timing_enable(2)  # the three stage kernels, so that the stand-alone map kernel runs
    starts = generic_starts(n)
    for e in (a, b):
        e.reset(np.arange(n), np.arange(n), starts=starts)
    for s, act in enumerate(SCRIPT + [(2, 0, 0), (0, 0, 0.9), (2, 0, 0)] * 5):
        odom = torch.tensor([act] * n, dtype=torch.float64, device=a.device)
        was = a.L.drlgx_debug_map_form(0)
        a.step(odom)
        a.synchronize()
        a.L.drlgx_debug_map_form(1)
        b.step(odom)
        b.synchronize()
        a.L.drlgx_debug_map_form(was)
        assert a.status() == 0 and b.status() == 0
        for i in range(n):
            pa, ia, ta, ua = a.virtual_map(i)
            pb, ib, tb, ub = b.virtual_map(i)
            np.testing.assert_array_equal(ua, ub)
            np.testing.assert_array_equal(pa, pb)
            np.testing.assert_array_equal(ia, ib)
            np.testing.assert_array_equal(ta, tb)
        np.testing.assert_array_equal(a.utility().cpu().numpy(), b.utility().cpu().numpy())
        np.testing.assert_array_equal(a.uncertainty_em(1).cpu().numpy(), b.uncertainty_em(1).cpu().numpy())
        np.testing.assert_array_equal(a.explored().cpu().numpy(), b.explored().cpu().numpy())
    a.close()
    b.close()


def test_state_struct_through_a_pointer_equals_by_value(monkeypatch):
    """The fused step reads the engine's state struct through a pointer to a device-resident copy (csrc/k_step.hip k_step_ref,
    kept current by drlgx_engine.cpp state_sync) instead of taking it by value in the kernel arguments; DRLGX_STATE_PTR=0 at
    creation keeps the by-value kernel.  Same code on the same struct: every state bit-equal, step after step - also across a
    call that CHANGES the struct between steps (the phase-stamp switch), which the copy has to follow."""
    n = 6
    monkeypatch.setenv("DRLGX_STATE_PTR", "0")
    a, cfg = make_engine(n, num_landmarks=60)
    monkeypatch.delenv("DRLGX_STATE_PTR")
    b, _ = make_engine(n, num_landmarks=60)
    starts = generic_starts(n)
    for e in (a, b):
        e.reset(np.arange(n), np.arange(n), starts=starts)
    import ctypes as C
    for s, act in enumerate(SCRIPT + [(2, 0, 0), (0, 0, 0.9), (2, 0, 0)] * 4):
        odom = torch.tensor([act] * n, dtype=torch.float64, device=a.device)
        if s == 5:  # the struct changes (a member is set): the device copy must be re-uploaded before the next launch
            for e in (a, b):
                e._chk(e.L.drlgx_debug_phase_clocks_host(e.h, 1, None))
        if s == 8:
            out = (C.c_int64 * 64)()
            for e in (a, b):
                e._chk(e.L.drlgx_debug_phase_clocks_host(e.h, 0, out))
        a.step(odom)
        b.step(odom)
        assert a.status() == 0 and b.status() == 0
        for i in range(n):
            for x, y in zip(a.poses(i) + a.landmarks(i) + a.virtual_map(i), b.poses(i) + b.landmarks(i) + b.virtual_map(i)):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a.utility().cpu().numpy(), b.utility().cpu().numpy())
    a.close()
    b.close()


def _staged_episode(e, seed, start, n_moves):
    """reset; measure + add at pose 0 WITHOUT an optimise there; then moves, each followed by the step's two measure calls,
    add and optimise (the staged C ABI, include/drlgx.h drlgx_stage_*)."""
    e.stage_reset([0], [seed], np.array([start]))
    k, br, c = e.stage_measure()
    e.stage_add_measurements(k, br, c)
    for _ in range(n_moves):
        e.stage_move(torch.tensor([[1.0, 1.0, math.pi / 2]], dtype=torch.float64, device=e.device))
        e.stage_measure()
        k, br, c = e.stage_measure()
        e.stage_add_measurements(k, br, c)
        e.stage_optimize()
    e.check_status()


def test_staged_reset_never_continues_the_previous_episodes_covariance_panel():
    """k_reset invalidates the incremental path's covariance panel (csrc/k_sim.hip): a staged reset runs no solve, so an
    episode whose last panel was left at one pose must not pass `inc_precheck` at the next episode's second pose.  The
    re-used engine must give, bit for bit, what a fresh engine gives for the second episode, by a FULL solve."""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    cfg = default_config(MAP, num_landmarks=100, max_poses=41, max_landmarks=100)
    used, fresh = Engine(cfg, 1, 0), Engine(cfg, 1, 0)
    starts = generic_starts(4)
    # episode 1 on `used`: fused reset = prior + measure + optimise at ONE pose -> panel (valid, P = 1, L, M) left behind
    used.reset([0], [3], starts=starts[3:4])
    used.synchronize()
    used.inc_stats(reset=True)
    for e in (used, fresh):
        _staged_episode(e, 1, starts[1], 1)
    assert used.inc_stats() == (0, 1), "the first optimise after a staged reset must be a full solve"
    for a, b in zip(used.poses(0) + used.landmarks(0) + used.factors(0), fresh.poses(0) + fresh.landmarks(0) + fresh.factors(0)):
        np.testing.assert_array_equal(a, b)
    # ... and twice in a row without any optimise at one pose in between
    for e in (used, fresh):
        e.stage_reset([0], [2], np.array([starts[2]]))
        _staged_episode(e, 1, starts[1], 3)
    for a, b in zip(used.poses(0) + used.landmarks(0), fresh.poses(0) + fresh.landmarks(0)):
        np.testing.assert_array_equal(a, b)
    for e in (used, fresh):
        e.close()


@pytest.mark.parametrize("num_landmarks, max_poses, steps", [(100, 64, 52), (8, 64, 48), (8, 41, 30), (300, 72, 66)])
def test_incremental_update_equals_the_full_solve_beyond_the_dense_solver_and_with_the_panel_in_lds(monkeypatch, num_landmarks, max_poses, steps):
    """The incremental path against DRLGX_INCREMENTAL=0 where the explicit 31-step test does not reach: trajectories beyond
    42 poses (the pose-chain solver builds the panel itself: k_slam_arrow.hip, mk_panel) and few-landmark worlds whose panel
    is LDS-resident for the step (k_inc.hip, kLds = true).  Estimates 1e-9, information 1e-7 relative between the engines
    and against the oracle at the end; `inc_stats` proves which path served the updates.  (300 landmarks in the 40 m world: beyond
    53 poses every step re-observes 20-40 landmarks with the covariance panel in HBM / L2 - the streamed form of the update,
    k_inc.hip: sbatch, with two to four factor tiles per walk and second batches.)"""
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    n = 4
    cfg = default_config(MAP, num_landmarks=num_landmarks, max_poses=max_poses, max_landmarks=min(num_landmarks, 127),
                         max_factors=None if num_landmarks <= 100 else 45 * max_poses)
    eng = Engine(cfg, n, 0)
    monkeypatch.setenv("DRLGX_INCREMENTAL", "0")
    ref = Engine(cfg, n, 0)
    monkeypatch.delenv("DRLGX_INCREMENTAL")
    assert ref.inc_stats() == (-1, -1)
    starts = generic_starts(n)
    ocfg = O.default_config(MAP, num_landmarks=num_landmarks)
    sims = [O.OracleSim(ocfg, lo, lo, start=tuple(starts[lo])) for lo in range(n)]
    for e in (eng, ref):
        e.reset(np.arange(n), np.arange(n), starts=starts)
    script = ([(1, 1, math.pi / 2)] * 4 + [(2, 0, 0), (1.5, 0, 0), (0, 0, 0.7), (2, 0, 0), (0, 0, 0.9)] * 12)[:steps]
    for s, act in enumerate(script):
        odom = torch.tensor([act] * n, dtype=torch.float64, device=eng.device)
        eng.step(odom)
        ref.step(odom)
        for sim in sims:
            sim.simulate(act)
        if s % 6 != 5 and s != len(script) - 1:
            continue
        assert eng.status() == 0 and ref.status() == 0
        for i in range(n):
            name = "step %d env %d" % (s, i)
            xyt, info = eng.poses(i)
            rxyt, rinfo = ref.poses(i)
            np.testing.assert_allclose(xyt, rxyt, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(info, rinfo, rtol=1e-7, atol=1e-6, err_msg=name)
            k, xy, linfo = eng.landmarks(i)
            rk, rxy, rlinfo = ref.landmarks(i)
            np.testing.assert_array_equal(k, rk)
            np.testing.assert_allclose(xy, rxy, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(linfo, rlinfo, rtol=1e-7, atol=1e-6, err_msg=name)
            np.testing.assert_array_equal(eng.virtual_map(i)[3], ref.virtual_map(i)[3])
            np.testing.assert_allclose(eng.virtual_map(i)[1], ref.virtual_map(i)[1], rtol=1e-6, atol=1e-9, err_msg=name)
            # the engine whose every update is a full solve against the oracle where the dense solver's sweep runs nine (43 .. 47
            # poses) and ten (48 .. 53) tile rows - two tile rows per wave (k_slam.hip: SweepRow)
            if s in (41, 47) and max_poses >= 54:
                compare_state(ref, i, sims[i], "full-solve engine, " + name, check_vm=False)
    for i in range(n):
        compare_state(eng, i, sims[i], "final env %d" % i, check_vm=False)
    inc, full = eng.inc_stats()
    assert inc + full == n * (len(script) + 1)
    assert inc >= n * (len(script) - len(script) // 10 - 4), (inc, full)  # all but the reset and the 10th updates
    eng.close()
    ref.close()


def test_more_envs_than_compute_units_run_the_map_stage_two_per_cu():
    """With more envs than the device has CUs `drlgx_step` keeps the map stage out of the fused kernel and launches it as the
    two-workgroups-per-CU kernel (csrc/drlgx_engine.cpp: drlgx_step, k_map_c).  Instances are independent and both map forms
    are bit-equal, so env i of a 300-env engine must equal, bit for bit, the same env stepped in a small engine."""
    n_big, pick = 300, [0, 1, 150, 298, 299]
    big, cfg = make_engine(n_big, num_landmarks=60)
    small, _ = make_engine(len(pick), num_landmarks=60)
    assert torch.cuda.get_device_properties(0).multi_processor_count < n_big
    rng = np.random.RandomState(5)
    starts = np.stack([rng.uniform(-8, 8, n_big), rng.uniform(-8, 8, n_big), rng.uniform(-3, 3, n_big)], 1)
    big.reset(np.arange(n_big), np.arange(n_big) + 7, starts=starts)
    small.reset(np.arange(len(pick)), np.array(pick) + 7, starts=starts[pick])
    sim = O.OracleSim(O.default_config(MAP, num_landmarks=60), pick[2] + 7, 0, start=tuple(starts[pick[2]]))
    for s, act in enumerate(SCRIPT + [(2, 0, 0), (0, 0, 0.9), (2, 0, 0)] * 3):
        big.step(torch.tensor([act] * n_big, dtype=torch.float64, device=big.device))
        small.step(torch.tensor([act] * len(pick), dtype=torch.float64, device=small.device))
        sim.simulate(act)
    assert big.status() == 0 and small.status() == 0
    for k, i in enumerate(pick):
        for a, b in zip(big.poses(i) + big.landmarks(i) + big.virtual_map(i), small.poses(k) + small.landmarks(k) + small.virtual_map(k)):
            np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(big.utility().cpu().numpy()[pick], small.utility().cpu().numpy())
    compare_state(big, pick[2], sim, "env %d of %d" % (pick[2], n_big))
    big.close()
    small.close()
