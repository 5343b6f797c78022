"""Dev tool: in-kernel phase breakdown of k_slam / k_sim at the bench workload:  phase_profile.py [workgroup = env index, default 0]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
if os.environ.get('DRLGX_LIB_DEV'):
    from drl_graph_exploration_amd.engine import Engine
    Engine.check_status = lambda self: None  # kernel-variant timing experiments produce wrong numerics on purpose
eng, cfg = bench.make_engine(0, 0)
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
out = (C.c_int64 * 64)()
BLK = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ARM = 1 | (BLK << 8)
eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, None)
print('workgroup %d:' % BLK, eng.counts(BLK) if hasattr(eng, 'counts') else '')
acc = np.zeros(64); n = 0
for it in range(20):
    eng.restore(0); eng.step(odom)
    eng.L.drlgx_debug_phase_clocks_host(eng.h, ARM, out)
    a = np.array(out[:], dtype=np.float64)
    if it >= 5:
        acc += a; n += 1
a = acc / n
names = {0:"start",1:"(front if not overlapped) + new factors",2:"landmark blocks + lists",3:"G",4:"Schur",5:"sweeps",6:"landmark partials",7:"outputs"}
print("k_slam phases after the simulator (us, block 0; the front end runs beside the simulator wave in k_step):")
for k in range(1, 8): print("  %-24s %8.2f" % (names[k], (a[k]-a[k-1]) / 100.0))
print("  total %.2f" % ((a[7]-a[0]) / 100.0))
print("simulator start -> SLAM back start: %.2f us; -> end of the SLAM front end (first front thread): %.2f us" % ((a[0] - a[8]) / 100.0, (a[14] - a[8]) / 100.0))
print("SLAM front end beside the simulator (us after the simulator's start): staged %.2f, factor tables %.2f, list starts %.2f; block assembly ends per wave 1..7: %s; barrier after both: %.2f" % ((a[33]-a[8])/100.0, (a[34]-a[8])/100.0, (a[35]-a[8])/100.0, " ".join("%.2f" % ((a[24+w]-a[8])/100.0) for w in range(1, 8)), (a[32]-a[8])/100.0))
print("k_sim phases (us, block 0): load %.2f, move %.2f, measure-1 %.2f, measure-2 %.2f, store %.2f, total %.2f" % (tuple((a[i+1]-a[i])/100.0 for i in (8,9,10,11,12)) + ((a[13]-a[8])/100.0,)))
seq = [(40, "table stores + clears"), (41, "landmark cells + pose LLT"), (17, "bbox"), (42, "mask clear"), (43, "range/FOV tests + compaction"),
       (21, "push-through"), (19, "cell pass"), (20, "block reduction")]
prev = a[16]
print("map stage inside k_step (us; the stand-alone kernel: scripts/phase_profile_map.py): SLAM end -> map start %.2f" % ((a[16] - a[7]) / 100.0))
for k, name in seq:
    print("  %-32s %7.2f" % (name, (a[k] - prev) / 100.0)); prev = a[k]
print("  cell pass per wave: end at +" + " ".join("%.2f" % ((a[48 + w] - a[21]) / 100.0) for w in range(8)))
print("  total %.2f; simulator start -> end of the map stage %.2f" % ((a[20] - a[16]) / 100.0, (a[20] - a[8]) / 100.0))
eng.timing_enable(True); eng.timing_read()
for it in range(50):
    eng.restore(0); eng.step(odom)
print(eng.timing_read())
