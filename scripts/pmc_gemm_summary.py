"""Summarise a rocprofv3 --pmc rocpd database per kernel (avg counter values per dispatch).
usage: python scripts/pmc_gemm_summary.py <results.db> [kernel-name substring]"""
import sqlite3
import sys

db = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "counters_collection"]
if not view:
    print("tables:", tabs)
    sys.exit(1)
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("# columns:", cols)
q = ("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection "
     "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name")
for r in cur.execute(q, ("%" + pat + "%",)):
    print("%-60s %-28s n=%-5d avg=%-14.1f max=%.1f" % (r[0][:60], r[1], r[2], r[3], r[4]))
extra = [c for c in ("grid_size", "workgroup_size", "lds_block_size", "vgpr_count", "accum_vgpr_count", "scratch_size", "sgpr_count") if c in cols]
if extra:
    q = "select kernel_name, %s, count(*) from counters_collection where kernel_name like ? group by kernel_name, grid_size" % ", ".join(extra)
    for r in cur.execute(q, ("%" + pat + "%",)):
        print(r[0][:50], dict(zip(extra, r[1:-1])), "n=", r[-1])
