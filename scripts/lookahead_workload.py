"""Dev tool: the bench's look-ahead workload alone (2 633 candidates from the 37-pose snapshot, 19 475 rollout updates), for
rocprofv3 passes: python scripts/lookahead_workload.py [repetitions = 6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
kmax = int(nact.max())
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.lookahead(cand_env, acts, nact, kmax)
torch.cuda.synchronize()
eng.check_status()
print("candidates %d, rollout updates %d" % (cand_env.numel(), int(nact.sum())))
