"""Dump the kernel statistics of a rocprofv3 (rocpd sqlite) result as CSV: name,calls,total_us,avg_us,percent."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print('"%s",%d,%.3f,%.3f,%.3f' % (name, calls, total, avg, pct))
