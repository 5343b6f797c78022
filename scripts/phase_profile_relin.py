"""Dev tool: in-kernel phases of a belief update that RELINEARISES (full dense solve + the covariance panel it leaves,
csrc/k_slam.hip + k_inc.hip: panel_from_dense) at the bench workload: the envs are stepped from the snapshot until the next
update is a 10th one, then that update is profiled per workgroup.   phase_profile_relin.py [workgroup ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
eng, cfg = bench.make_engine(0, 0, max_poses=int(os.environ.get("PP_CAP", "41")))
odom = torch.tensor([bench.STEP_ACTION] * bench.N_ENVS, dtype=torch.float64, device=eng.device)
eng.restore(0)
while (eng.counts(0)["isam_count"] + 1) % 10 != 0:
    eng.step(odom)
extra = int(os.environ.get("PP_EXTRA_TENS", "0"))
for _ in range(10 * extra):
    eng.step(odom)
eng.check_status()
eng.snapshot(0)
c = eng.counts_dev().cpu().numpy()
print("update #%d: poses %d -> %d, landmarks mean %.1f max %d" % (eng.counts(0)["isam_count"] + 1, c[0, 0], c[0, 0] + 1, c[:, 1].mean(), c[:, 1].max()))
eng.inc_stats(True)
eng.step(odom)
print("served incrementally / by a full solve:", eng.inc_stats(True))
names = [(0, "start"), (1, "front + new factors"), (2, "landmark blocks + lists"), (3, "G"), (4, "Schur"), (5, "sweep"), (6, "landmark partials"),
         (7, "outputs"), (44, "(panel starts)"), (45, "panel: pose rows (Sigma_pl)"), (46, "panel: landmark rows (Sigma_ll)")]
out = (C.c_int64 * 64)()
blocks = [int(v) for v in sys.argv[1:]] or [0, int(np.argmax(c[:, 1])), int(np.argmin(c[:, 1]))]
for per_stage in (2, 1):
    eng.timing_enable(per_stage)
    for blk in blocks:
        arm = 1 | (blk << 8)
        acc = np.zeros(64); n = 0
        for it in range(8):
            eng.restore(0)
            eng.L.drlgx_debug_phase_clocks_host(eng.h, arm, None)
            eng.step(odom)
            eng.L.drlgx_debug_phase_clocks_host(eng.h, arm, out)
            if it >= 3:
                acc += np.array(out[:], dtype=np.float64); n += 1
        a = acc / n
        print("%s, workgroup %d (%d landmarks): " % ("stage kernel k_slam" if per_stage == 2 else "fused k_step", blk, c[blk, 1]) +
              ", ".join("%s %.1f" % (nm, (a[k] - a[names[i - 1][0]]) / 100.0) for i, (k, nm) in enumerate(names) if i) +
              "; total %.1f us" % ((a[46] - a[0]) / 100.0))
    eng.timing_read()
    for it in range(6):
        eng.restore(0); eng.step(odom)
    print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in eng.timing_read().items() if v[1]})
eng.close()
