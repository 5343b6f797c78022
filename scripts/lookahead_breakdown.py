"""Dev tool: where the look-ahead time goes (engine span timers) on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
eng, cfg = bench.make_engine(0, 0, max_poses=64)
dev = eng.device
eng.restore(0)
g = eng.graph()
nfr = g["n_frontier"].long()
cand_env = torch.repeat_interleave(torch.arange(bench.N_ENVS, device=dev), nfr).to(torch.int32)
first = torch.cumsum(nfr, 0) - nfr
fidx = torch.arange(cand_env.numel(), device=dev) - first[cand_env.long()]
goals = g["frontier_xy"][cand_env.long(), fidx].contiguous()
acts, nact = eng.line_plan(cand_env, goals)
na = nact.cpu().numpy()
print("candidates %d, actions per plan: mean %.2f max %d; total rollout steps %d; histogram %s" % (len(na), na.mean(), na.max(), na.sum(), np.bincount(na)))
eng.lookahead(cand_env, acts, nact)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.lookahead(cand_env, acts, nact)
torch.cuda.synchronize()
print("lookahead %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
eng.inc_stats(True)
eng.lookahead(cand_env, acts, nact)
print("one look-ahead: SLAM updates served incrementally / by a full solve:", eng.inc_stats(True))
eng.lookahead(cand_env, acts, nact, int(na.max()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.lookahead(cand_env, acts, nact, int(na.max()))
torch.cuda.synchronize()
print("lookahead bounded by the longest plan %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
eng.timing_enable(True); eng.timing_read()
for _ in range(5):
    eng.lookahead(cand_env, acts, nact)
tm = eng.timing_read()
print({k: (round(v[0] / 5, 3), v[1] // 5) for k, v in tm.items() if v[1]}); eng.check_status()
# one bounded look-ahead by span, and its cost per action index: bounded calls of increasing depth (the difference of two
# consecutive depths is what one action index's launches cost)
mx = int(na.max())
eng.timing_read()
eng.lookahead(cand_env, acts, nact, mx)
print("spans of one bounded look-ahead (ms, launches):", {k: (round(v[0], 3), v[1]) for k, v in eng.timing_read().items() if v[1]})
eng.timing_enable(False)
prev = 0.0
for a in range(1, mx + 1):
    eng.lookahead(cand_env, acts, nact, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.lookahead(cand_env, acts, nact, a)
    torch.cuda.synchronize()
    ms_a = (time.perf_counter() - t0) / 3 * 1e3
    print("  action indices < %2d: %.3f ms (+%.3f); rollouts running at the last one: %d" % (a, ms_a, ms_a - prev, int((na >= a).sum())))
    prev = ms_a
