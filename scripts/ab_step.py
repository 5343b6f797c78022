"""Dev tool: the bench's timed loop (restore + step) and the look-ahead round, for A/B runs of a kernel-variant library
(DRLGX_LIB_DEV=<library>) against the product one on the same box: ab_step.py [steps = 300]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
eng.timing_enable(True)
for _ in range(30):
    eng.restore(0); eng.step(odom)
eng.timing_read()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.restore(0); eng.step(odom)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
t = eng.timing_read()
print("%-60s step %.2f us  (k_step %.2f us, copy %.2f us per launch)" % (os.environ.get("DRLGX_LIB_DEV", "product")[-60:], dt * 1e6,
      t["step"][0] / max(t["step"][1], 1) * 1e3, t["copy"][0] / max(t["copy"][1], 1) * 1e3))
eng.close()
