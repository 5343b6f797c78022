"""Dev tool: start / end stamps of every workgroup of the fused step kernel at the bench workload (how much of the kernel's
duration is dispatch ramp, how much the slowest instance)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
out = (C.c_int64 * 1024)()
eng.L.drlgx_debug_phase_clocks_host(eng.h, 1, None)
rows = []
for it in range(14):
    eng.restore(0); eng.step(odom)
    eng.L.drlgx_debug_phase_clocks_host(eng.h, 5, C.cast(out, C.POINTER(C.c_int64)))
    a = np.array(out[:], dtype=np.int64)
    if it >= 4:
        rows.append(a[128:128 + 2 * bench.N_ENVS].reshape(-1, 2).copy())
r = np.stack(rows).astype(np.float64) / 100.0  # us
start = r[:, :, 0] - r[:, :, 0].min(axis=1, keepdims=True)
dur = r[:, :, 1] - r[:, :, 0]
span = (r[:, :, 1].max(axis=1) - r[:, :, 0].min(axis=1))
print("workgroup start offsets (us): mean %.2f, p50 %.2f, max %.2f" % (start.mean(), np.median(start), start.max(axis=1).mean()))
print("workgroup durations (us): min %.2f, mean %.2f, p90 %.2f, max %.2f; block 0 %.2f" % (dur.min(axis=1).mean(), dur.mean(), np.percentile(dur, 90, axis=1).mean(), dur.max(axis=1).mean(), dur[:, 0].mean()))
print("first start -> last end (us): %.2f" % span.mean())
cnts = [eng.counts(i) for i in range(bench.N_ENVS)]
L = np.array([c["landmarks"] for c in cnts]); M = np.array([c["factors"] for c in cnts])
d = dur.mean(axis=0)
print("corr(duration, landmarks) %.2f, corr(duration, factors) %.2f; landmarks min/mean/max %d/%.1f/%d; factors %d/%.1f/%d" % (np.corrcoef(d, L)[0, 1], np.corrcoef(d, M)[0, 1], L.min(), L.mean(), L.max(), M.min(), M.mean(), M.max()))
order = np.argsort(d)
print("slowest 5 workgroups:", [(int(i), round(float(d[i]), 1), int(L[i]), int(M[i])) for i in order[-5:]])
for lo_, hi_ in ((0, 18), (19, 26), (27, 100)):
    sel_ = (L >= lo_) & (L <= hi_)
    if sel_.any():
        print("landmarks %2d..%3d: %3d workgroups, mean duration %.1f us (factors mean %.0f)" % (lo_, hi_, sel_.sum(), d[sel_].mean(), M[sel_].mean()))
