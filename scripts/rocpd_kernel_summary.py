"""Dev tool: per-kernel totals of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes DIR/NAME_results.db):
python scripts/rocpd_kernel_summary.py DB [top_n] [divide_by]  ->  total busy time, span, and the top kernels (ms, launches, mean us)."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = list(db.execute("select name, start, end from kernels order by start"))
tot, cnt = collections.Counter(), collections.Counter()
for n, s, e in rows:
    k = n[:90]
    tot[k] += e - s
    cnt[k] += 1
print("%d launches, busy %.2f ms (/%g = %.3f ms), span %.1f ms" % (len(rows), sum(tot.values()) / 1e6, div, sum(tot.values()) / 1e6 / div,
                                                                  (rows[-1][2] - rows[0][1]) / 1e6))
for n, v in tot.most_common(top):
    print("%9.2f ms %6d x %8.1f us  %s" % (v / 1e6, cnt[n], v / cnt[n] / 1e3, n))
if len(sys.argv) > 4:  # by launch count
    print("-- by launches")
    for n, c in cnt.most_common(int(sys.argv[4])):
        print("%6d x %8.1f us (%.1f per unit)  %s" % (c, tot[n] / c / 1e3, c / div, n))
