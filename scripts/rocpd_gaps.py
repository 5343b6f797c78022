"""Dev tool: device-idle gaps of a rocprofv3 rocpd database: python scripts/rocpd_gaps.py DB [min_us] [skip_launches] -> per (kernel before,
kernel after) pair: count, total idle ms - which host sections leave the device waiting."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(db.execute("select name, start, end from kernels order by start"))[skip:]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:46]
tot, cnt = collections.Counter(), collections.Counter()
idle = 0
end = rows[0][2]
for i in range(1, len(rows)):
    g = rows[i][1] - end
    if g > min_us * 1e3:
        k = (short(rows[i - 1][0]), short(rows[i][0]))
        tot[k] += g; cnt[k] += 1
    if g > 0: idle += g
    end = max(end, rows[i][2])
print("span %.1f ms, idle %.1f ms" % ((rows[-1][2] - rows[0][1]) / 1e6, idle / 1e6))
for k, v in tot.most_common(25):
    print("%8.2f ms %4d x %7.1f us   %s  ->  %s" % (v / 1e6, cnt[k], v / cnt[k] / 1e3, k[0], k[1]))
