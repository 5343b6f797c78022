"""Dev tool: in-kernel phases of ONE rollout's last incremental step inside the look-ahead's loop kernel (bench workload, plans cut to
K actions so that the stamped action is an incremental one): phase_profile_lookahead.py [rollout = 0] [K = 3]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
BLK = int(sys.argv[1]) if len(sys.argv) > 1 else 0
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nact = torch.clamp(nact, max=K)
out = (C.c_int64 * 64)()
ARM = 1 | (BLK << 8)
eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, None)
acc = np.zeros(64); n = 0
for it in range(12):
    eng.lookahead(cand_env, acts, nact, K)
    eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, out)
    if it >= 4:
        acc += np.array(out[:], dtype=np.float64); n += 1
a = acc / n
us = lambda i, j: (a[i] - a[j]) / 100.0
print("rollout %d (env %d), action %d of %d:" % (BLK, int(cand_env[BLK]), K - 1, K))
print("  first half: loads (state + panel -> LDS) %.2f, new pose rows %.2f; -> barrier %.2f" % (us(1, 0), us(2, 1), us(32, 2)))
print("  factor lists + linearisation %.2f; measurement update %.2f; new landmarks %.2f; outputs %.2f; panel write-back + meta %.2f" % (
    us(35, 32), us(3, 35), us(4, 3), us(5, 4), us(7, 5)))
print("  first half start -> end of write-back %.2f" % us(7, 0))
