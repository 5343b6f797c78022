"""Dev tool: durations (us) of the launches whose kernel name contains PATTERN, in launch order, REPS consecutive ones averaged:
python scripts/rocpd_by_order.py DB PATTERN [REPS]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",))]
print(" ".join("%.0f" % (min(d[i:i + reps])) for i in range(0, len(d), reps)))
