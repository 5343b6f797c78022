"""Dev tool: a longer A/B of the look-ahead / plan-execution paths than the test suite runs - the loop kernels with the simulator
run ahead (default), the loop kernels with the simulator inside every step (DRLGX_LOOKAHEAD_PRESIM=0) and one launch per action
index (DRLGX_LOOKAHEAD_LOOP=0) must give bit-equal rewards and states.   ab_lookahead_paths.py [envs = 64] [decisions = 30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drl_graph_exploration_amd.vecenv import VecExplorationEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D = int(sys.argv[2]) if len(sys.argv) > 2 else 30
num_lm = int(sys.argv[3]) if len(sys.argv) > 3 else None
envs = []
for env_var in (None, "DRLGX_LOOKAHEAD_PRESIM", "DRLGX_LOOKAHEAD_LOOP"):
    if env_var:
        os.environ[env_var] = "0"
    kw = dict(num_landmarks=num_lm) if num_lm else {}
    envs.append(VecExplorationEnv(40, n, env_index=7, test=True, device=0, **kw))
    if env_var:
        del os.environ[env_var]
n_cand = n_resets = 0
for d in range(D):
    raws = []
    for e in envs:
        e.graph_matrix()
        e.actions_all_goals()
        raws.append(e.rewards_all_goals(return_raw=True)[1])
    if not (torch.equal(raws[0], raws[1]) and torch.equal(raws[0], raws[2])):
        e = envs[0]
        c = e.engine.counts_dev().cpu().numpy()
        for k, name in ((1, "presim off"), (2, "loop off")):
            bad = torch.nonzero(raws[0] != raws[k]).view(-1).cpu().numpy()
            print("decision %d, default vs %s: %d of %d candidates differ" % (d, name, len(bad), raws[0].numel()))
            for b in bad[:8]:
                env = int(e._cand_env[b])
                print("   cand %d env %d (poses %d landmarks %d isam %d) n_act %d: %.17g vs %.17g" % (b, env, c[env, 0], c[env, 1], c[env, 4], int(e._n_act[b]), float(raws[0][b]), float(raws[k][b])))
        k12 = torch.nonzero(raws[1] != raws[2]).numel()
        print("presim-off vs loop-off differ in %d" % k12)
        sys.exit(1)
    n_cand += raws[0].numel()
    nfr = envs[0]._graph["n_frontier"].long()
    choice = (torch.arange(n, device=envs[0].device) * 3 + d) % nfr
    dones = []
    for e in envs:
        _, done, _ = e.step(choice)
        dones.append(done | e.truncated())
    assert torch.equal(dones[0], dones[1]) and torch.equal(dones[0], dones[2])
    m = [e.metrics() for e in envs]
    assert torch.equal(m[0], m[1]) and torch.equal(m[0], m[2]), "metrics differ at decision %d" % d
    if bool(dones[0].any()):
        ids = np.nonzero(dones[0].cpu().numpy())[0]
        n_resets += len(ids)
        for e in envs:
            e.reset(ids)
c = envs[0].engine.counts_dev().cpu().numpy()
for i in range(0, n, max(1, n // 8)):
    for x, y, z in zip(*[e.engine.poses(i) + e.engine.landmarks(i) + e.engine.virtual_map(i) for e in envs]):
        assert np.array_equal(x, y) and np.array_equal(x, z)
print("%d envs x %d decisions: %d candidates' rewards, metrics and sampled states bit-equal across the three paths; %d episode resets; "
      "poses now mean %.0f max %d, landmarks max %d" % (n, D, n_cand, n_resets, c[:, 0].mean(), c[:, 0].max(), c[:, 1].max()))
