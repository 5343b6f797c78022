"""Dev tool: in-kernel phase breakdown (one workgroup) of k_map at the bench workload, as a stage kernel.
usage: phase_profile_map.py [n_envs = 256] [workgroup = 0]    (DRLGX_MAP_COMPACT=0/1 picks the kernel form)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
NE = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ENVS
BLK = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bench.N_ENVS = NE
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * NE, dtype=torch.float64, device=eng.device)
ARM = 1 | (BLK << 8)
out = (C.c_int64 * 64)()
eng.timing_enable(2)
eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, None)
acc = np.zeros(64); n = 0
for it in range(25):
    eng.restore(0); eng.step(odom)
    eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, out)
    a = np.array(out[:], dtype=np.float64)
    if it >= 5:
        acc += a; n += 1
a = acc / n
seq = [(40, "stage loads"), (41, "landmark cells + pose LLT"), (17, "bbox"), (42, "mask clear"), (43, "range/FOV tests + compaction"),
       (21, "push-through")]
prev = a[16]
print("k_map phases (us, workgroup %d of %d, DRLGX_MAP_COMPACT=%s):" % (BLK, NE, os.environ.get("DRLGX_MAP_COMPACT", "-")))
for k, name in seq:
    print("  %-32s %7.2f" % (name, (a[k] - prev) / 100.0)); prev = a[k]
ends = [(a[48 + w] - a[21]) / 100.0 for w in range(8)]
print("  phase C per wave: end at +" + " ".join("%.2f" % e for e in ends))
print("  phase C per wave: inside CI loops " + " ".join("%.2f" % (a[56 + w] / 100.0) for w in range(8)))
print("  phase C (to the barrier)         %7.2f" % ((a[19] - a[21]) / 100.0))
print("  block reduction                  %7.2f" % ((a[20] - a[19]) / 100.0))
print("  total                            %7.2f" % ((a[20] - a[16]) / 100.0))
print(eng.timing_read())
