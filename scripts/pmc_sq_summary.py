"""Per-kernel averages of the SQ / GRBM counters of one rocprofv3 --pmc pass (rocpd sqlite), for the belief-step kernels,
with the derived ratios DESIGN.md quotes (per dispatch):
  VALU issue share  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES          (both in quad-cycles, summed over waves)
  wait share        = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  MFMA pipe busy    = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), SIMDs = 4 x number of workgroups (one per CU)
usage: pmc_sq_summary.py <results.db> [<results2.db> ...] [--json out.json]
(several passes - the SQ block has 8 counter slots - are merged per kernel; --json also writes the averages with the
digest of the kernel sources, which bench.py checks before quoting them in `roofline_issue`)"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNELS = ("k_step", "k_slam_arrow", "k_slam", "k_map", "k_sim_step", "k_copy_instances")
argv = sys.argv[1:]
json_out = None
if "--json" in argv:
    json_out = argv[argv.index("--json") + 1]
    del argv[argv.index("--json"):argv.index("--json") + 2]
rows = []
for db in argv:
    cur = sqlite3.connect(db).cursor()
    rows += list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                             "group by kernel_name, counter_name"))
by = {}
for name, ctr, n, avg in rows:
    short = next((k for k in KERNELS if k in name), None)
    if short:
        by.setdefault(short, {})[ctr] = (n, avg)
for k in KERNELS:
    if k not in by:
        continue
    c = by[k]
    print("%s  (%d dispatches)" % (k, next(iter(c.values()))[0]))
    for ctr in sorted(c):
        print("    %-28s %16.1f" % (ctr, c[ctr][1]))
    g = lambda x: c[x][1] if x in c else None  # noqa: E731
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_ACTIVE_INST_VALU") is not None:
            print("    VALU issue share of wave cycles      %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
        if g("SQ_WAIT_INST_ANY") is not None:
            print("    issue-stall share (WAIT_INST_ANY)    %.3f" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
        if g("SQ_WAIT_ANY") is not None:
            print("    parked share (WAIT_ANY: waitcnt/barrier) %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
        if g("SQ_ACTIVE_INST_ANY") is not None:
            print("    issuing share (ACTIVE_INST_ANY)      %.3f" % (g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print("    VALU instructions per wave           %.0f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        print("    MFMA busy cycles / GRBM_GUI_ACTIVE   %.3f  (divide by the SIMDs in use for a pipe utilisation)" %
              (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("GRBM_GUI_ACTIVE")))
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_INSTS_LDS"):
        print("    LDS bank-conflict cycles per LDS instruction  %.2f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")))

if json_out:
    import bench  # noqa: E402  (csrc_digest)
    with open(json_out, "w") as f:
        json.dump({"csrc_sha1": bench.csrc_digest(),
                   "source": "rocprofv3 --pmc SQ_* (two passes of 8 SQ counters + GRBM_GUI_ACTIVE) on `python bench.py --steps 50 "
                             "--warmup 5 --no-cpu-baseline --no-policy --no-train`; averages per dispatch",
                   "kernels": {k: {c: v[1] for c, v in by[k].items()} for k in by}}, f, indent=1)
