"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately, rocpd .db output) into
per-kernel HBM traffic per launch.  Usage: rocpd_pmc.py <fetch.db> <write.db> > profiles/rNN_pmc_traffic.json

Units/corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B
read requests at 64 B, so it is doubled (the instance-copy kernel, whose byte count is known exactly - it reads what it
writes - confirms the factor on this workload: 2 x 12860 KiB read vs 25244 KiB written)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    q = ("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
         "group by kernel_name")
    return {r[0]: (r[1], r[2]) for r in cur.execute(q, (counter,))}


def short(name):
    for k in ("k_step_arrow_loop", "k_step_loop", "k_step_arrow", "k_step", "k_slam_arrow", "k_slam", "k_map_c", "k_map", "k_sim_step",
              "k_copy_instances", "k_reset"):
        if k in name:
            return k
    return name


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name in fetch:
    if name not in write:
        continue
    f, w = fetch[name][1] * 1024.0, write[name][1] * 1024.0
    out[short(name)] = {"launches_sampled": fetch[name][0], "FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes": w,
                        "hbm_read_bytes_corrected": 2.0 * f, "hbm_traffic_bytes_per_launch": 2.0 * f + w}
import bench  # noqa: E402  (csrc_digest: ties the figures to the kernel sources they were measured on)
print(json.dumps({"csrc_sha1": bench.csrc_digest(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                            + (sys.argv[3] if len(sys.argv) > 3 else "`python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-policy`"),
                  "correction": "FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": out}, indent=1))
