"""GPU parity tests of the reference-shaped host layer: `VecExplorationEnv` against the oracle's ExplorationEnv
restatement driven with the same decisions, the single-env `EMExplorer` facade, and a short `DeepQ.running`."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

MAP = 40


def test_vec_env_follows_oracle_env_over_decisions():
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    n = 5
    starts = np.array([O.start_pose(lo, MAP / 2 + 20) for lo in range(n)]) + np.array([0.2871, -0.3179, 0.0917])
    # 7 rollout instances < number of (env, frontier) candidates: the look-ahead runs in several waves
    env = VecExplorationEnv(MAP, n, env_index=0, test=True, starts=starts, max_poses=60, n_rollouts=7)
    refs = [O.OracleEnv(MAP, lo, start=tuple(starts[lo])) for lo in range(n)]
    assert [int(s) for s in env.env_index] == [r.env_index for r in refs]
    # Dead-reckoned estimates sit on "round" coordinates and line plans run along 3-4-5 directions towards odd-integer
    # cell centres, so cells at EXACTLY max_range from a pose occur structurally; 1 ulp of pose round-off decides them
    # (in the reference too). An env whose grids differ ONLY in such knife-edge cells is dropped from the comparison
    # from then on (its frontier set, hence its decisions, legitimately fork); any other difference fails.
    alive = [True] * n
    for decision in range(6):
        g = env.graph_matrix()
        env.actions_all_goals()
        rew, raw = env.rewards_all_goals(return_raw=True)
        cand_env, cand_node, first = env.candidates
        rew_h, raw_h, first_h = rew.cpu().numpy(), raw.cpu().numpy(), first.cpu().numpy()
        nfr = g["n_frontier"].cpu().numpy()
        node_off = g["node_off"].cpu().numpy()
        choice = np.zeros(n, dtype=np.int64)
        plans = []
        for i, r in enumerate(refs):
            if not alive[i]:
                plans.append([])
                continue
            A, X, _, fro = r.graph_matrix()
            assert fro == nfr[i] and A.shape[0] == node_off[i + 1] - node_off[i]
            acts = r.actions_all_goals()
            exp = r.rewards_all_goals(acts)
            ks = A.shape[0] - fro
            got = rew_h[first_h[i]:first_h[i] + fro]
            # the normalisation divides by (max - min) of rewards that agree to ~1e-7
            np.testing.assert_allclose(got, exp[ks:], atol=5e-5)
            assert bool(env.loop_clo[i]) == r.loop_clo
            assert int(cand_node[first_h[i]]) - node_off[i] == ks
            choice[i] = int(np.argmax(exp[ks:])) if decision % 2 == 0 else decision % fro
            plans.append(acts[ks + choice[i]])
        _, done, _ = env.step(choice)
        for i, r in enumerate(refs):
            for a in plans[i]:
                r.step(a)
        ex = env.status().cpu().numpy()
        dist = env.dist.cpu().numpy()
        for i, r in enumerate(refs):
            if not alive[i]:
                continue
            pe, po = env.obs(i), r._sim.virtual_map()[0]
            if not np.array_equal(pe, po):
                knife = r._sim.knife_edge_cells(1e-9).reshape(pe.shape)
                assert knife[pe != po].all(), "grids differ outside knife-edge cells"
                alive[i] = False
                continue
            assert ex[i] == r.status()
            assert dist[i] == pytest.approx(r.dist, abs=1e-12)
            # done() = the reference's condition; "within one plan of the engine's pose capacity" is reported apart
            near_full = r._sim.num_poses() + env.cfg.max_actions + 1 > env.cfg.max_poses
            assert bool(done[i]) == r.done()
            assert bool(env.truncated()[i]) == near_full
    assert sum(alive) >= 3
    k = alive.index(True)
    assert env.get_landmark_error(k) == pytest.approx(refs[k].get_landmark_error(), abs=1e-6)
    assert env.max_uncertainty_of_trajectory(k) == pytest.approx(refs[k].max_uncertainty_of_trajectory(), rel=1e-5)
    env.close()


def test_emexplorer_facade_single_env():
    from configparser import ConfigParser
    from drl_graph_exploration_amd.pyplanner2d import EMExplorer
    cp = ConfigParser()
    cp.read_dict({
        "Sensor Model": dict(bearing_noise="0.5", range_noise="0.02", min_bearing="-179.9", max_bearing="179.9",
                             min_range="0.1", max_range="6.0"),
        "Control Model": dict(translation_noise="0.1", rotation_noise="0.2"),
        "Environment": dict(min_x="-20", max_x="20", min_y="-20", max_y="20", max_steps="5000", safe_distance="0.0"),
        "Virtual Map": dict(resolution="2.0", sigma0="1.0", num_samples="1"),
        "Simulator": dict(seed="3", lo="3", num="8", sigma_x0="0.05", sigma_y0="0.05", sigma_theta0="0.01"),
        "Planner": dict(seed="3", angle_weight="0.4", distance_weight0="5.0", distance_weight1="2.0", d_weight="0.0",
                        max_edge_length="2.0", max_nodes="0.5", occupancy_threshold="0.4", safe_distance="1.0",
                        algorithm="EM_AOPT", reg_out="false"),
    })
    start = tuple(np.array(O.start_pose(3, MAP / 2 + 20)) + np.array([0.2871, -0.3179, 0.0917]))
    sim = EMExplorer(cp, start=start)
    ref = O.OracleSim(O.default_config(MAP), 3, 3, start=start)
    for _ in range(4):
        assert sim.simulate((1, 1, math.pi / 2)) is False
        ref.simulate((1, 1, math.pi / 2))
    assert sim.step == ref.step
    assert sim._slam.map.get_landmark_size() == ref.num_landmarks()
    assert sim._slam.key_size() == ref.key_size()
    # integer start poses leave knife-edge cells (range == max_range up to round-off): compare the others
    mask = ~ref.knife_edge_cells().reshape(ref.vm_shape())
    np.testing.assert_array_equal(sim._virtual_map.to_array()[mask], ref.virtual_map()[0][mask])
    veh = sim.vehicle_position
    np.testing.assert_allclose([veh.x, veh.y, veh.theta], ref.poses()[0][-1], atol=1e-7)
    assert sim.calculate_utility(1.5) == pytest.approx(ref.calculate_utility(1.5), rel=1e-6)
    sim._slam.adjacency_degree_get()
    A, X = ref.adjacency()
    np.testing.assert_allclose(sim._slam.adjacency_out(), A, atol=1e-7)
    np.testing.assert_allclose(sim._slam.features_out()[:, 0], X, rtol=1e-5, atol=1e-9)
    goal = (veh.x + 3.0, veh.y - 2.0)
    plan = sim.line_plan(0, goal)
    oplan = ref.line_plan(goal)
    assert len(plan) == len(oplan)
    np.testing.assert_allclose([[a.x, a.y, a.theta] for a in plan], oplan, atol=1e-9)
    assert sim.simulations_reward(plan) == pytest.approx(ref.simulations_reward(oplan), abs=1e-6)
    assert sim.simulate((100.0, 0.0, 0.0)) is True  # rejected odometry, state untouched
    assert sim.step == ref.step


def test_plan_execution_with_one_map_rebuild_equals_rebuilding_every_action():
    """`step` executes a plan action by action (exploration_env.py:98-105).  The virtual map is a pure function of the SLAM
    state, so drlgx_step_plan rebuilds it (and the marginals it needs) at each env's last action only: the state after the
    plan - estimates, information, virtual map, metrics - is bit-equal to rebuilding after every action, over two decisions."""
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    n = 6
    envs = [VecExplorationEnv(40, n, env_index=3, test=True, device=0) for _ in range(2)]
    for d in range(2):
        choice = None
        for e, every in zip(envs, (True, False)):
            e.graph_matrix()
            acts, nact = e.actions_all_goals()
            if choice is None:
                nfr = e._graph["n_frontier"].long()
                choice = (torch.arange(n, device=e.device) * 3 + d) % nfr  # some frontier of every env
                assert int(nact[e._cand_first + choice].max()) >= 3     # multi-action plans
            c = e._cand_first + choice
            e.step_actions(acts[c], nact[c], map_every_action=every)
        a, b = envs
        assert torch.equal(a.metrics(), b.metrics())
        assert torch.equal(a.status(), b.status())
        for i in range(n):
            for x, y in zip(a.engine.poses(i), b.engine.poses(i)):
                np.testing.assert_array_equal(x, y)
            for x, y in zip(a.engine.landmarks(i), b.engine.landmarks(i)):
                np.testing.assert_array_equal(x, y)
            for x, y in zip(a.engine.virtual_map(i), b.engine.virtual_map(i)):
                np.testing.assert_array_equal(x, y)
    for e in envs:
        e.close()


def test_long_plans_with_one_map_rebuild_stay_within_tolerance():
    """The same property beyond the LDS-resident solver (43+ poses: pose-chain solver, whose solve-only steps sum the pose
    deltas in another order than the full ones): scripted 8-action plans up to ~80 poses, the state after every plan within
    the engine-vs-oracle tolerances of the state rebuilt after every action."""
    import math
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    n = 4
    starts = np.array([[-7.3183, -6.2718, 0.1234], [3.1, 4.7, 2.2], [-2.4, 8.8, -1.0], [9.1, -3.3, 0.5]])
    envs = [VecExplorationEnv(40, n, env_index=0, test=True, device=0, starts=starts, num_landmarks=30) for _ in range(2)]
    loop = [(2, 0, 0)] * 3 + [(0, 0, math.pi / 2)] + [(2, 0, 0)] * 2 + [(0.7, 0, 0.4), (1.0, 0, -0.3)]
    dev = envs[0].device
    acts = torch.zeros(n, envs[0].cfg.max_actions, 3, dtype=torch.float64, device=dev)
    acts[:, :len(loop)] = torch.tensor(loop, dtype=torch.float64, device=dev)
    nact = torch.tensor([8, 7, 8, 5], dtype=torch.int32, device=dev)
    for plan in range(10):
        for e, every in zip(envs, (True, False)):
            e.step_actions(acts, nact, map_every_action=every)
        a, b = envs
        np.testing.assert_allclose(b.metrics().cpu().numpy(), a.metrics().cpu().numpy(), rtol=1e-7)
        for i in range(n):
            (pa, ia), (pb, ib) = a.engine.poses(i), b.engine.poses(i)
            assert pa.shape == pb.shape
            np.testing.assert_allclose(pb, pa, rtol=0, atol=1e-9)
            np.testing.assert_allclose(ib, ia, rtol=1e-7, atol=1e-9 * np.abs(ia).max())
            va, vb = a.engine.virtual_map(i), b.engine.virtual_map(i)
            np.testing.assert_array_equal(vb[0], va[0])  # occupancy probabilities: ladder states
            for x, y in zip(va[1:], vb[1:]):
                np.testing.assert_allclose(y, x, rtol=1e-7, atol=1e-12)
    assert int(envs[0].engine.counts(0)["poses"]) > 80
    for e in envs:
        e.close()


def test_deepq_running_smoke(tmp_path):
    from drl_graph_exploration_amd.networks import GCN
    from drl_graph_exploration_amd.policy import DeepQ
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    dq = DeepQ("smoke/", "GCN", data_root=str(tmp_path))
    dq.OBSERVE, dq.epoch, dq.BATCH = 16, 48, 16
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    w0 = pol.conv2.weight.detach().clone()
    dq.running(pol, tgt, test=True, n_envs=8)
    assert dq.step_t == 48 and len(dq.buffer) == 48
    assert dq.temp_loss > 0 and math.isfinite(dq.temp_loss)
    assert not torch.equal(w0, pol.conv2.weight.detach())  # at least one Adam step on the HIP backward
    assert (tmp_path / "training_object_data" / "smoke" / "Model_Policy.pt").exists()


def test_deepq_running_at_256_envs(tmp_path):
    """BASELINE configs[2] at its own size: `DeepQ.running` over 256 lock-step environments with the reference's mini-batch of
    64 graphs and one update per environment step (3 vector steps: one to fill the replay, two with 256 updates each)."""
    from drl_graph_exploration_amd.networks import GCN
    from drl_graph_exploration_amd.policy import DeepQ
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    dq = DeepQ("smoke256/", "GCN", data_root=str(tmp_path))
    dq.OBSERVE, dq.epoch = 256, 768
    assert dq.BATCH == 64
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    w0 = pol.conv2.weight.detach().clone()
    dq.running(pol, tgt, test=True, n_envs=256)
    assert dq.step_t == 768 and len(dq.buffer) == 768
    assert dq.temp_loss > 0 and math.isfinite(dq.temp_loss)
    assert torch.isfinite(pol.conv2.weight).all() and not torch.equal(w0, pol.conv2.weight.detach())


def test_deepq_reload_repools_the_replay_buffer(tmp_path):
    """run_training.py re-loads saved_training.pkl between epochs (train.py:85-94): the pickled replay buffer carries its
    graphs as host tensors.  The next epoch puts them back into the device pool (DeepQ._repool), so that its updates take
    the pooled collation and the cached target read-out like the ones before the reload - and give the same mini-batches:
    a graph read back through its PoolRef equals the host graph it was made from."""
    import pickle
    from drl_graph_exploration_amd.networks import GCN, PoolRef
    from drl_graph_exploration_amd.policy import DeepQ
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    dq = DeepQ("reload/", "GCN", data_root=str(tmp_path))
    dq.OBSERVE, dq.epoch, dq.BATCH = 16, 48, 16
    pol, tgt = GCN().to(dev), GCN().to(dev)
    tgt.load_state_dict(pol.state_dict())
    dq.running(pol, tgt, test=True, n_envs=8)
    dq2 = pickle.loads(pickle.dumps(dq))
    assert len(dq2.buffer) == 48 and not any(isinstance(t[0], PoolRef) or isinstance(t[3], PoolRef) for t in dq2.buffer)
    host = [(t[0].x.clone(), t[0].edge_index.clone(), t[0].edge_attr.clone(), t[3].x.clone(), t[3].edge_index.clone()) for t in dq2.buffer]
    from drl_graph_exploration_amd.networks import ReplayPool
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    env = VecExplorationEnv(40, 8, env_index=0, test=True, device=0, seed=0)
    mn, me, _ = env.engine.graph_capacity()
    pool = ReplayPool(dev, 20, mn, me)
    dq2._repool(pool, dev)
    assert all(isinstance(t[0], PoolRef) and isinstance(t[3], PoolRef) for t in dq2.buffer)
    for t, h in zip(dq2.buffer, host):
        assert torch.equal(t[0].x.cpu(), h[0]) and torch.equal(t[0].edge_index.cpu(), h[1]) and torch.equal(t[0].edge_attr.cpu(), h[2])
        assert torch.equal(t[3].x.cpu(), h[3]) and torch.equal(t[3].edge_index.cpu(), h[4])
    env.close()
    # ... and the trainer carries on from the pickle (its own pool, its own re-pooling)
    dq3 = pickle.loads(pickle.dumps(dq))
    dq3.epoch = 24
    dq3.running(pol, tgt, test=True, n_envs=8)
    assert dq3.step_t == 72 and all(isinstance(t[0], PoolRef) for t in dq3.buffer)
    assert dq3.temp_loss > 0 and math.isfinite(dq3.temp_loss)


def test_a2c_running_smoke(tmp_path):
    """A2C (scripts/policy.py:262-503) over the vectorised env: n-step returns per env, actor + critic on the HIP GCN,
    one optimiser step per `nstep` vector steps."""
    from drl_graph_exploration_amd.networks import PolicyGCN, ValueGCN
    from drl_graph_exploration_amd.policy import A2C
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    a2c = A2C("smoke_a2c/", data_root=str(tmp_path))
    a2c.nstep, a2c.epoch, a2c.graphs_per_pass = 3, 24, 5  # 6 vector steps of 4 envs -> 2 updates, 3 chunks each
    actor, critic = PolicyGCN().to(dev), ValueGCN().to(dev)
    wa, wc = actor.conv2.weight.detach().clone(), critic.fully_con1.weight.detach().clone()
    a2c.running(actor, critic, test=True, n_envs=4)
    assert a2c.step_t == 24 and len(a2c.buffer) == 0
    assert math.isfinite(a2c.temp_loss) and a2c.temp_loss != 0 and a2c.entro > 0
    assert not torch.equal(wa, actor.conv2.weight.detach()) and not torch.equal(wc, critic.fully_con1.weight.detach())
    for f in ("Model_Policy.pt", "Model_Value.pt", "temp_loss.csv"):
        assert (tmp_path / "training_object_data" / "smoke_a2c" / f).exists()
    rows = (tmp_path / "reward_data" / "smoke_a2c" / "reward_data.csv").read_text().strip().splitlines()
    assert rows[0] == "Step,Reward" and len(rows) == 25


def test_a2c_resumes_from_a_pickle_taken_mid_window(tmp_path):
    """run_training.py re-loads saved_training.pkl between epochs (train.py:85-94).  An epoch rarely ends on a multiple of
    nstep, so the pickled trainer carries a partial n-step window whose graphs are host copies: the next epoch must move
    them back to the device and train on the mixed window (pooled + carried-over graphs)."""
    import pickle
    from drl_graph_exploration_amd.networks import PolicyGCN, ValueGCN
    from drl_graph_exploration_amd.policy import A2C
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    a2c = A2C("resume_a2c/", data_root=str(tmp_path))
    a2c.nstep, a2c.epoch, a2c.graphs_per_pass = 3, 16, 5  # 4 vector steps of 4 envs: one update, ONE step left in the window
    actor, critic = PolicyGCN().to(dev), ValueGCN().to(dev)
    a2c.running(actor, critic, test=True, n_envs=4)
    assert a2c.step_t == 16 and len(a2c.buffer) == 1
    again = pickle.loads(pickle.dumps(a2c))
    assert len(again.buffer) == 1 and all(not d.x.is_cuda for d in again.buffer[0][0])
    again.temp_loss = 0.0
    again.running(actor, critic, test=True, n_envs=4)  # 4 more vector steps: the window fills after two of them
    assert again.step_t == 32 and len(again.buffer) == 2
    assert math.isfinite(again.temp_loss) and again.temp_loss != 0


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own golden data through the PRODUCT path: data/test_result/40_DQN_GCN.csv (fixture csv_pin.json)
# ---------------------------------------------------------------------------------------------------------------------
def test_hip_path_reproduces_reference_evaluation_csv(golden_dir):
    """Every pinned row of the reference's shipped evaluation run (data/test_result/40_DQN_GCN.csv: per executed action
    the landmark error, the map entropy and the largest pose-covariance trace; 2 273 rows over 44 seeds, one whole
    179-action episode, five seeds beyond 90 actions) replayed through the product path in ONE vectorised run:
    VecExplorationEnv (HIP belief step with the pose-chain solver beyond 42 poses, device metrics, graph export, line
    plans) and the HIP GCN with the shipped MyModel.pt.

    The engine executes the action lists the fixture pins (the reference's actions as inferred by the CPU oracle's
    search, scripts/make_csv_pin_fixture.py), so the rows do not depend on knife-edge decisions of the replaying
    implementation (a frontier tie, a cell at exactly max_range, a path length of exactly 4.0 m).  Tolerances as in the
    oracle's CPU pin (tests/test_oracle_csv_pin.py): 1e-4 relative on landmark error and trace, 2e-2 on the entropy.
    Beside that the product's own decisions are compared where they are well defined: its line plan to the pinned goal
    (equal to the executed plan up to the documented remainder variants), its frontier list (contains the pinned goal)
    and the HIP GCN's pick (the reference network's, on the decisions where the frontier lists agree)."""
    import json
    import os
    from drl_graph_exploration_amd.networks import GCN, GraphData
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    pins = json.load(open(os.path.join(golden_dir, "csv_pin.json")))["seeds"]
    seeds = [int(k) for k, v in pins.items() if len(v["rows"]) >= 4]
    n = len(seeds)
    assert n >= 40
    env = VecExplorationEnv(MAP, n, env_index=0, test=True, max_poses=256, n_rollouts=0)
    env.env_index = np.array(seeds, dtype=np.int64)
    env.reset()
    dev = env.device
    A_max = env.cfg.max_actions
    model = GCN()
    model.load_state_dict(torch.load(os.path.join(golden_dir, "DQN_GCN_MyModel.pt"), map_location="cpu"))
    model.to(dev)
    row = [0] * n
    bad_rows = []
    plan_same = plan_all = goal_in_frontier = gcn_same = gcn_all = 0
    n_dec = max(len(pins[str(s)]["choices"]) for s in seeds)
    for d in range(n_dec):
        g = env.graph_matrix()
        with torch.no_grad():
            q = model(GraphData(g["x"], g["edge_index"], g["edge_attr"], g["batch"]), 0.0, batch=g["batch"]).view(-1).cpu().numpy()
        node_off, nfr = g["node_off"].cpu().numpy(), g["n_frontier"].cpu().numpy()
        fxy = g["frontier_xy"].cpu().numpy()
        live = np.array([d < len(pins[str(s)]["choices"]) for s in seeds])
        goals = np.zeros((n, 2))
        acts = np.zeros((n, A_max, 3))
        nact = np.zeros(n, dtype=np.int64)
        for i, s in enumerate(seeds):
            if not live[i]:
                continue
            pin = pins[str(s)]
            goals[i] = pin["goals"][d]
            pl = np.array(pin["plans"][d])
            acts[i, :len(pl)] = pl
            nact[i] = len(pl)
            ch = pin["choices"][d]
            fr = fxy[i, :nfr[i]]
            hit = np.nonzero(np.all(np.abs(fr - goals[i]) < 1e-9, axis=1))[0]
            goal_in_frontier += int(len(hit) > 0)
            if isinstance(ch, int) and len(hit) and hit[0] == ch and pin["gcn_choices"][d] >= 0:
                gcn_all += 1
                gcn_same += int(np.argmax(q[node_off[i + 1] - nfr[i]:node_off[i + 1]])) == pin["gcn_choices"][d]
        # the product's own line plan to the pinned goals
        ce = torch.arange(n, dtype=torch.int32, device=dev)
        pa, pn = env.engine.line_plan(ce, torch.as_tensor(goals, device=dev))
        pa, pn = pa.cpu().numpy(), pn.cpu().numpy()
        for i, s in enumerate(seeds):
            if not live[i]:
                continue
            want = acts[i, :nact[i]]
            mine = pa[i, :pn[i]]
            plan_all += 1
            last_cut = pins[str(s)]["finished"] and d + 1 == len(pins[str(s)]["choices"])  # the reference stopped mid-plan
            if last_cut:
                plan_same += int(len(want) <= len(mine) and np.allclose(mine[:len(want)], want, atol=1e-7))
                continue
            # equal up to the side of the remainder when the path length is a multiple of the edge length to one ulp
            # (int(d / 2) full edges + remainder, Planner2D.cpp:1027-1036): same rotations, same total length, and the
            # action counts differ by at most the zero-length / full-length tail
            rot_m, rot_w = mine[mine[:, 0] == 0], want[want[:, 0] == 0]
            nz_m, nz_w = mine[np.abs(mine[:, 0]) > 1e-9], want[np.abs(want[:, 0]) > 1e-9]
            plan_same += int(abs(len(mine) - len(want)) <= 1 and abs(mine[:, 0].sum() - want[:, 0].sum()) < 1e-7 and
                             abs(mine[:, 2].sum() - want[:, 2].sum()) < 1e-7 and len(nz_m) == len(nz_w))
        a_dev = torch.as_tensor(acts, device=dev)
        na = torch.as_tensor(nact, device=dev)
        for k in range(int(nact.max())):
            active = (na > k).to(torch.uint8)
            env.engine.step(a_dev[:, k].contiguous(), active)
            m = env.engine.metrics().cpu().numpy()
            for i, s in enumerate(seeds):
                if nact[i] <= k:
                    continue
                ref = np.array(pins[str(s)]["rows"][row[i]])
                rel = np.abs(m[i] - ref) / np.abs(ref)
                if rel[0] > 1e-4 or rel[2] > 1e-4 or rel[1] > 2e-2:
                    bad_rows.append((s, row[i], rel.tolist()))
                row[i] += 1
        env._graph = None
    env.engine.check_status()
    poses = env.engine.counts_dev()[:, 0].cpu().numpy()
    env.close()
    total = sum(len(pins[str(s)]["rows"]) for s in seeds)
    assert sum(row) == total and total >= 2000
    assert poses.max() >= 180  # the whole 179-action episode ran on the device
    assert not bad_rows, bad_rows[:5]
    assert plan_same >= 0.97 * plan_all, (plan_same, plan_all)
    assert goal_in_frontier >= 0.85 * plan_all, (goal_in_frontier, plan_all)
    assert gcn_all >= 300 and gcn_same >= 0.97 * gcn_all, (gcn_same, gcn_all)


def test_free_running_product_reproduces_reference_rows(golden_dir):
    """The same golden rows WITHOUT the fixture's plans: on the seeds whose pinned decisions are all the reference network's
    greedy picks with the plain line plan (no frontier override, no remainder variant, no goal search - six seeds, 204 rows,
    of which 166 are reached before a documented knife edge),
    VecExplorationEnv + the HIP GCN make their OWN decisions - graph export, Q values, arg-max over the frontier nodes, line
    plan to that frontier, execution - and must land on every pinned row (scripts/test.py:104-142 is exactly this loop).

    The CPU oracle runs beside each env on the product's actions; the two occupancy maps must agree in every cell that is
    not a knife-edge cell (a cell centre at exactly max_range of a dead-reckoned pose: which side it falls on is decided by
    the last digit of an executed action - the product's line plans equal the pinned ones to ~1e-15 - and by floating-point
    noise in the reference itself; oracle.knife_edge_cells).  A seed leaves the comparison only if a DECISION differs from the
    reference's while its map holds such cells, or if a plan differs from the reference's by a zero-length tail action (a
    path of exactly two edge lengths); every decision before that, and every other seed to its last pinned row, must agree."""
    import json
    import os
    from drl_graph_exploration_amd.networks import GCN, GraphData
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    pins = json.load(open(os.path.join(golden_dir, "csv_pin.json")))["seeds"]
    seeds = [int(k) for k, v in pins.items() if len(v["rows"]) >= 4 and
             all(isinstance(c, int) and c == g for c, g in zip(v["choices"], v["gcn_choices"]))]
    n = len(seeds)
    assert n >= 5 and sum(len(pins[str(s)]["rows"]) for s in seeds) >= 200
    env = VecExplorationEnv(MAP, n, env_index=0, test=True, max_poses=256, n_rollouts=0)
    env.env_index = np.array(seeds, dtype=np.int64)
    env.reset()
    refs = [O.OracleEnv(MAP, s) for s in seeds]
    dev = env.device
    model = GCN()
    model.load_state_dict(torch.load(os.path.join(golden_dir, "DQN_GCN_MyModel.pt"), map_location="cpu"))
    model.to(dev)
    row = [0] * n
    want_rows = [len(pins[str(s)]["rows"]) for s in seeds]
    bad_rows, wrong, diverged, knife_seen, tail = [], [], {}, set(), {}
    for d in range(max(len(pins[str(s)]["choices"]) for s in seeds)):
        for i, s in enumerate(seeds):  # knife-edge events of the reference itself
            if i in diverged or row[i] >= want_rows[i]:
                continue
            pe, po = env.engine.virtual_map(i)[0].reshape(-1), refs[i]._sim.virtual_map()[0].reshape(-1)
            knife = refs[i]._sim.knife_edge_cells(1e-9)
            assert not np.any((pe != po) & ~knife), "a cell that is not a knife-edge cell differs (seed %d, decision %d)" % (s, d)
            if knife.any():  # (the last digit of an executed action decides which side such a cell falls on)
                knife_seen.add(i)
        g = env.graph_matrix()
        with torch.no_grad():
            q = model(GraphData(g["x"], g["edge_index"], g["edge_attr"], g["batch"]), 0.0, batch=g["batch"]).view(-1).cpu().numpy()
        node_off, nfr = g["node_off"].cpu().numpy(), g["n_frontier"].cpu().numpy()
        fxy = g["frontier_xy"].cpu().numpy()
        live = np.array([row[i] < want_rows[i] and d < len(pins[str(s)]["choices"]) and i not in diverged and i not in tail for i, s in enumerate(seeds)])
        goals = np.zeros((n, 2))
        for i, s in enumerate(seeds):
            if live[i]:
                pick = int(np.argmax(q[node_off[i + 1] - nfr[i]:node_off[i + 1]]))
                goals[i] = fxy[i, pick]
                if not np.all(np.abs(goals[i] - np.array(pins[str(s)]["goals"][d])) < 1e-9):
                    if i in knife_seen:  # (knife-edge cells in its map: its own trajectory from here)
                        diverged[i] = d
                        live[i] = False
                    else:
                        wrong.append((s, d, pick, pins[str(s)]["choices"][d], goals[i].tolist(), pins[str(s)]["goals"][d]))
        pa, pn = env.engine.line_plan(torch.arange(n, dtype=torch.int32, device=dev), torch.as_tensor(goals, device=dev))
        pn = pn.cpu().numpy() * live
        pa_h = pa.cpu().numpy()
        for i, s in enumerate(seeds):
            # the other documented knife edge: a path of exactly two edge lengths, int(d / 2) full edges + remainder
            # (Planner2D.cpp:1027-1036) - the plans then differ by a zero-length tail action and nothing else
            if live[i]:
                want, mine = np.array(pins[str(s)]["plans"][d]), pa_h[i, :pn[i]]
                if len(want) != len(mine) or not np.allclose(mine, want, atol=1e-7):
                    k0 = min(len(want), len(mine))
                    assert abs(len(want) - len(mine)) == 1 and np.allclose(mine[:k0], want[:k0], atol=1e-7) and \
                        np.allclose((want if len(want) > k0 else mine)[k0:], 0.0, atol=1e-7), (s, d, want, mine)
                    tail[i] = d
        na = torch.as_tensor(pn, device=dev)
        for k in range(int(pn.max())):
            env.engine.step(pa[:, k].contiguous(), (na > k).to(torch.uint8))
            m = env.engine.metrics().cpu().numpy()
            for i, s in enumerate(seeds):
                if pn[i] <= k:
                    continue
                refs[i].step(tuple(pa_h[i, k]))
                if row[i] >= want_rows[i]:
                    continue
                ref = np.array(pins[str(s)]["rows"][row[i]])
                rel = np.abs(m[i] - ref) / np.abs(ref)
                if rel[0] > 1e-4 or rel[2] > 1e-4 or rel[1] > 2e-2:
                    bad_rows.append((s, row[i], rel.tolist()))
                row[i] += 1
        env._graph = None
    env.engine.check_status()
    env.close()
    assert not wrong, wrong
    assert not bad_rows, bad_rows[:5]
    done = [i for i in range(n) if row[i] == want_rows[i]]
    # seeds 9, 36, 40 run to their last pinned row; 42 leaves with knife-edge cells different (its 15th decision), 23 that
    # way or at a zero-length plan tail (7th / 8th decision), 20 at a zero-length tail (4th)
    assert len(done) + len(diverged) + len(tail) == n and len(done) >= 3 and len(diverged) + len(tail) <= 3, \
        (row, want_rows, diverged, tail)
    assert sum(row) >= 155, (row, want_rows)


def test_device_metrics_equal_the_host_getters():
    """drlgx_metrics (landmark error, map entropy of scripts/test.py:61-74, max pose-covariance trace) against the same
    quantities assembled on the host from the exported state, and against the oracle."""
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    n = 5
    env = VecExplorationEnv(MAP, n, env_index=3, test=True, max_poses=60, n_rollouts=0)
    refs = [O.OracleEnv(MAP, 3 + i) for i in range(n)]
    for act in [(2.0, 0.0, 0.0), (0.0, 0.0, 1.1), (2.0, 0.0, 0.0), (1.5, 0.0, 0.0)]:
        env.engine.step(torch.tensor([act] * n, dtype=torch.float64, device=env.device))
        for r in refs:
            r.step(act)
    m = env.metrics().cpu().numpy()
    for i in range(n):
        obs = env.obs(i)
        host = [env.get_landmark_error(i), float(-(obs * np.log(obs)).sum() + 0.5 * np.log(0.5) * 1200),
                env.max_uncertainty_of_trajectory(i)]
        np.testing.assert_allclose(m[i], host, rtol=1e-12)
        orc = [refs[i].get_landmark_error(), O.map_entropy(refs[i]._obs), refs[i].max_uncertainty_of_trajectory()]
        # (the reference's integer start poses leave cells at exactly max_range: the entropy may differ by a few cells)
        np.testing.assert_allclose(m[i][[0, 2]], np.array(orc)[[0, 2]], rtol=1e-8)
        assert m[i][1] == pytest.approx(orc[1], rel=5e-3)
    ln, an = env.engine.cov_array()
    # VirtualMap.to_cov_array: untouched cells have covariance sigma0^2 I -> length sigma0; touched cells are tighter
    ln = ln.cpu().numpy()
    assert ln.shape == (n, 40, 40) and np.all(ln <= 1.0 + 1e-15) and (ln < 0.999).any() and np.isfinite(an.cpu().numpy()).all()
    env.close()


def test_train_scripts_write_the_reference_artefacts(tmp_path):
    """scripts/train.py + run_training.py over the HIP path: two short DQN epochs (one in-process, one as the reference's
    per-epoch subprocess) leave saved_training.pkl (trainer + replay), Model_Policy.pt / Model_Target.pt, temp_*.csv and a
    TensorBoard log with the reference's tags."""
    import os
    import pickle
    import subprocess
    import sys
    from drl_graph_exploration_amd import tfevents, train
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = str(tmp_path / "data")
    log = train.main(["DQN", "GCN", "--data-root", data, "--epochs", "1", "--epoch-steps", "16", "--observe", "8", "--batch", "8", "--n-envs", "4"])
    obj = os.path.join(data, "training_object_data", "DQN_GCN")
    for f in ("saved_training.pkl", "Model_Policy.pt", "Model_Target.pt", "temp_reward.csv", "temp_loss.csv"):
        assert os.path.exists(os.path.join(obj, f)), f
    with open(os.path.join(obj, "saved_training.pkl"), "rb") as f:
        tr = pickle.load(f)
    assert tr.step_t == 16 and len(tr.buffer) == 16 and tr.buffer[0][0].x.device.type == "cpu"
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.check_call([sys.executable, "-m", "drl_graph_exploration_amd.run_training", "DQN", "GCN", "--data-root", data, "--n-envs", "4"],
                          env=env, cwd=root, timeout=600)
    with open(os.path.join(obj, "saved_training.pkl"), "rb") as f:
        tr2 = pickle.load(f)
    assert tr2.step_t == 32 and len(tr2.buffer) == 32
    sc = tfevents.read_scalars(log)
    assert sc and {t for _, _, t, _ in sc} <= {"Train/avg_reward", "Train/loss"} and any(t == "Train/loss" for _, _, t, _ in sc)


def test_whole_plans_in_one_launch_equal_one_launch_per_action(monkeypatch):
    """Look-ahead rollouts and plan execution run a candidate's / an env's whole action list in ONE launch (csrc/k_step.hip:
    k_step_loop, k_step_arrow_loop - the fused step once per action inside the workgroup) when the fused kernels serve every
    pose count the plans can reach; DRLGX_LOOKAHEAD_LOOP=0 keeps one launch per action index.  Same kernels' bodies in the
    same order: rewards and the state after the plans bit-equal, over decisions that take the trajectories from the dense
    solver (<= 53 poses) to the pose-chain solver.  (The loop form keeps the per-action choice between the two solvers: they
    round differently, and the integer worlds hold cells at exactly max_range from a pose - with enough candidates some
    rollout's relinearising update falls where the two forms would otherwise pick different solvers, scripts/ab_lookahead_paths.py
    found that with 64 envs.)"""
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    n = 24
    a = VecExplorationEnv(40, n, env_index=2, test=True, device=0)
    monkeypatch.setenv("DRLGX_LOOKAHEAD_LOOP", "0")
    b = VecExplorationEnv(40, n, env_index=2, test=True, device=0)
    monkeypatch.delenv("DRLGX_LOOKAHEAD_LOOP")
    # ... and the rollouts' simulator run ahead for the whole action list and replayed (ksim::k_presim / replay_step_body, the
    # default) against simulating inside every rollout step (DRLGX_LOOKAHEAD_PRESIM=0): the same draws in the same order
    monkeypatch.setenv("DRLGX_LOOKAHEAD_PRESIM", "0")
    c = VecExplorationEnv(40, n, env_index=2, test=True, device=0)
    monkeypatch.delenv("DRLGX_LOOKAHEAD_PRESIM")
    for d in range(22):
        raws = []
        for e in (a, b, c):
            e.graph_matrix()
            e.actions_all_goals()
            raws.append(e.rewards_all_goals(return_raw=True)[1])
        assert torch.equal(raws[0], raws[1]) and torch.equal(raws[0], raws[2]), "decision %d" % d
        if d in (0, 7):
            # the rollouts themselves after a look-ahead: the replayed form (k_presim stores the simulator's own state once, behind the
            # last action) leaves the ground truth where simulating inside every step leaves it
            roll0 = 2 * n
            for k in (0, 1, 5, 17):
                np.testing.assert_array_equal(a.engine.ground_truth(roll0 + k)[0], c.engine.ground_truth(roll0 + k)[0])
                assert a.engine.counts(roll0 + k) == c.engine.counts(roll0 + k)
        nfr = a._graph["n_frontier"].long()
        choice = (torch.arange(n, device=a.device) * 2 + d) % nfr
        for e in (a, b, c):
            e.step(choice)
        assert torch.equal(a.metrics(), b.metrics()) and torch.equal(a.metrics(), c.metrics())
    for i in range(n):
        for x, y in zip(a.engine.poses(i) + a.engine.landmarks(i) + a.engine.virtual_map(i),
                        b.engine.poses(i) + b.engine.landmarks(i) + b.engine.virtual_map(i)):
            np.testing.assert_array_equal(x, y)
    # both solvers' regimes were visited, and trajectories beyond 64 poses: there the map stage works in pose chunks (one mask bit
    # per pose), and the loop kernels size the chunk for the range's last action while every action passes its own pose bound
    assert max(a.engine.counts(i)["poses"] for i in range(n)) > 66
    for e in (a, b, c):
        e.close()


def test_engine_fetch_brings_tensors_and_the_status_word_in_one_read():
    """Engine.fetch (drlgx_status_fetch_host): every dtype the trainers read comes back with the values and shapes of `.cpu()`, empty
    and unaligned pieces included; the same read refreshes the host's pose bounds and raises on a non-zero status word."""
    from drl_graph_exploration_amd import _lib
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    env = VecExplorationEnv(MAP, 3, env_index=0, test=True, max_poses=40)
    eng, dev = env.engine, env.device
    g = torch.Generator(device=dev).manual_seed(1)
    ts = [torch.randn(7, 3, device=dev, generator=g), torch.randn(5, device=dev, dtype=torch.float64, generator=g),
          torch.arange(11, device=dev, dtype=torch.int32), torch.arange(3, device=dev), torch.rand(9, device=dev, generator=g) > 0.5,
          torch.empty(0, device=dev), (torch.arange(13, device=dev) % 3).to(torch.uint8)[1:]]
    out = eng.fetch(*ts)
    assert len(out) == len(ts)
    for t, h in zip(ts, out):
        assert h.shape == tuple(t.shape) and np.array_equal(h, t.cpu().numpy())
    assert eng.fetch() == []  # (a plain status read)
    big = torch.arange(300000, device=dev, dtype=torch.float32)  # (beyond the first staging buffer: it grows)
    assert np.array_equal(eng.fetch(big)[0], big.cpu().numpy())
    # the counts' column of poses is what the engine selects its kernels by: exact after a read
    assert np.array_equal(eng.fetch(eng.counts_dev())[0][:, 0], np.array([eng.counts(i)["poses"] for i in range(3)]))
    # a capacity overflow (plans longer than the pose capacity) sets the status word: the next read raises
    acts = torch.zeros(3, env.cfg.max_actions, 3, dtype=torch.float64, device=dev)
    acts[:, :, 0] = 0.5
    nact = torch.full((3,), env.cfg.max_actions, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.DrlgxError):
        for _ in range(40):
            env.step_actions(acts, nact, kmax=env.cfg.max_actions, check=False)
        eng.fetch(nact)
    env.close()


def test_export_with_plans_equals_the_separate_calls():
    """graph_matrix(plan=True) - the line plans to every frontier slot computed inside the export's call and synchronisation - gives the
    candidates, plans, rewards and step of graph_matrix() + actions_all_goals(), bit for bit; a host array as the choice steps like a
    device tensor."""
    from drl_graph_exploration_amd.vecenv import VecExplorationEnv
    n = 6
    envs = [VecExplorationEnv(MAP, n, env_index=3, test=True, max_poses=80) for _ in range(2)]
    rng = np.random.RandomState(5)
    for decision in range(4):
        ga, gb = envs[0].graph_matrix(plan=True), envs[1].graph_matrix()
        assert envs[0]._n_act_h is not None and envs[1]._n_act_h is None
        for k in ("x", "edge_index", "edge_attr", "node_off", "edge_off", "n_frontier", "frontier_xy"):
            assert torch.equal(ga[k], gb[k]), k
        aa, na = envs[0].actions_all_goals()
        ab, nb = envs[1].actions_all_goals()
        assert torch.equal(na, nb) and np.array_equal(envs[0]._n_act_h, nb.cpu().numpy())
        live = torch.arange(aa.shape[1], device=aa.device)[None, :] < na[:, None]
        assert torch.equal(aa[live], ab[live])
        for x, y in zip(envs[0].candidates, envs[1].candidates):
            assert torch.equal(x, y)
        ra, rb = envs[0].rewards_all_goals(), envs[1].rewards_all_goals()
        assert torch.equal(ra, rb) and torch.equal(envs[0].loop_clo, envs[1].loop_clo)
        nfr = ga["n_frontier_h"]
        choice = np.array([rng.randint(0, max(int(f), 1)) for f in nfr], dtype=np.int64)
        _, da, _ = envs[0].step(choice, check=False)                                   # host choice: no synchronisation inside
        _, db, _ = envs[1].step(torch.as_tensor(choice, device=envs[1].device))      # device choice
        da_h, = envs[0].engine.fetch(da)
        assert np.array_equal(da_h, db.cpu().numpy())
        assert torch.equal(envs[0].engine.counts_dev(), envs[1].engine.counts_dev())
        assert torch.equal(envs[0].dist, envs[1].dist)
    for e in envs:
        e.close()
