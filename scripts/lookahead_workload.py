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
import time
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rew = None
for r in range(reps):
    if r == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    rew = eng.lookahead(cand_env, acts, nact, kmax)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / max(1, reps - 2) * 1e3 if reps > 2 else float("nan")
eng.check_status()
print("candidates %d, rollout updates %d; %.3f ms per look-ahead (host clock, %d repetitions); reward checksum %.17g" %
      (cand_env.numel(), int(nact.sum()), dt, max(0, reps - 2), float(rew.double().sum())))
if os.environ.get("LA_SPANS"):
    eng.timing_enable(True); eng.timing_read()
    for _ in range(5):
        eng.lookahead(cand_env, acts, nact, kmax)
    print("spans (ms, launches) per look-ahead:", {k: (round(v[0] / 5, 3), v[1] // 5) for k, v in eng.timing_read().items() if v[1]})
    eng.timing_enable(False)
