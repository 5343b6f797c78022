"""Dev tool: the head of a fused-step workgroup at the bench workload - how long after a workgroup's first stamp its simulator begins, has its
streams in LDS, finishes the move, ends: phase_profile_entry.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
out = (C.c_int64 * 1024)()
for BLK in (0, 141, 200):
    eng.L.drlgx_debug_phase_clocks_host(eng.h, 1 | (BLK << 8), None)
    acc = []
    for it in range(20):
        eng.restore(0); eng.step(odom)
        eng.L.drlgx_debug_phase_clocks_host(eng.h, 5 | (BLK << 8), C.cast(out, C.POINTER(C.c_int64)))
        a = np.array(out[:], dtype=np.float64)
        if it >= 5:
            st = a[128 + 2 * BLK]
            acc.append([(a[8] - st) / 100, (a[9] - st) / 100, (a[10] - st) / 100, (a[13] - st) / 100, (a[32] - st) / 100, (a[129 + 2 * BLK] - st) / 100,
                        (a[128:128 + 512:2].min() - st) / 100])
    m = np.mean(acc, axis=0)
    print("wg %d: after its start (us): simulator body begins %.2f, streams in LDS %.2f, move done %.2f, simulator ends %.2f, barrier %.2f, workgroup ends %.2f; first workgroup of the launch started %.2f before" % (BLK, *m[:6], -m[6]))
