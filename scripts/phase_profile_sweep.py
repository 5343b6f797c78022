"""Dev tool: per-wave cycle stamps of ONE block step (KI = 3) of k_slam's sweep (block 0) at the bench workload.
S.prof[64 + 5 w + k], k = 0 start, 1 after the panel publish, 2 before the second barrier, 3 after it, 4 end of the step."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
out = (C.c_int64 * 64)()
eng.L.drlgx_debug_phase_clocks_host(eng.h, 1, None)
rows = []
for it in range(12):
    eng.restore(0); eng.step(odom)
    eng.L.drlgx_debug_phase_clocks_host(eng.h, 3, out)
    a = np.array(out[:], dtype=np.int64)
    if it >= 4:
        rows.append(a[0:40].reshape(8, 5).copy())
r = np.stack(rows)
t0 = r[:, :, 0].min(axis=1)
rel = (r - t0[:, None, None]).mean(axis=0)
print("sweep block step KI=3, stamps relative to the earliest wave (clock64 ticks), mean of %d launches" % len(rows))
print("wave   start  published  E:mfma  E:inv-end    end   (columns 3, 4: the inverting wave only)")
for w in range(8):
    print("%4d %7.0f %9.0f %7.0f %7.0f %7.0f" % ((w,) + tuple(rel[w])))
