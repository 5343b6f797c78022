"""Dev tool: in-kernel phase breakdown of the incremental SLAM stage (csrc/k_inc.hip) inside k_step at the bench workload:
phase_profile_inc.py [workgroup = env index, default 0]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
if os.environ.get("DRLGX_LIB_DEV"):
    from drl_graph_exploration_amd.engine import Engine
    Engine.check_status = lambda self: None  # kernel-variant timing experiments produce wrong numerics on purpose
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
out = (C.c_int64 * 64)()
BLK = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ARM = 1 | (BLK << 8)
eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, None)
print('workgroup %d:' % BLK, eng.counts(BLK))
acc = np.zeros(64); n = 0
eng.inc_stats(True)
for it in range(20):
    eng.restore(0); eng.step(odom)
    eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, out)
    a = np.array(out[:], dtype=np.float64)
    if it >= 5:
        acc += a; n += 1
print("incremental / full updates:", eng.inc_stats())
a = acc / n
us = lambda i, j: (a[i] - a[j]) / 100.0
print("first half, beside the simulator (us after the simulator's start): begins %.2f, loads done %.2f, new pose done %.2f; simulator ends %.2f; barrier after both %.2f" % (
    us(0, 8), us(1, 8), us(2, 8), us(13, 8), us(32, 8)))
print("second half (us):")
print("  factor lists + linearisation          %.2f" % us(35, 32))
print("  B  measurement update (all batches)   %.2f   first batch: Y %.2f, T + inverse %.2f, tiles %.2f" % (us(3, 35), us(36, 35), us(37, 36), us(38, 37)))
print("  C  new landmarks                      %.2f" % us(4, 3))
print("  D  outputs                            %.2f" % us(5, 4))
print("     panel write-back, meta             %.2f" % us(7, 5))
print("  total %.2f; stage end -> map start %.2f; map stage %.2f; simulator start -> end of map %.2f" % (us(7, 32), us(16, 7), us(20, 16), us(20, 8)))
eng.timing_enable(True); eng.timing_read()
for it in range(50):
    eng.restore(0); eng.step(odom)
print(eng.timing_read())
