"""Dump the kernel statistics of a rocprofv3 (rocpd sqlite) result as CSV: name,calls,total_us,avg_us,percent.
With --by-grid [substring]: per (kernel, grid size) averages for the kernels whose name contains the substring."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
if len(sys.argv) > 2 and sys.argv[2] == "--by-grid":
    pat = sys.argv[3] if len(sys.argv) > 3 else ""
    print("kernel,grid_x,grid_y,grid_z,calls,avg_us,min_us")
    q = ("select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration) from kernels where name like ? "
         "group by name, grid_x, grid_y, grid_z order by name, grid_x")
    for name, gx, gy, gz, n, avg, mn in db.execute(q, ("%" + pat + "%",)):
        print('"%s",%d,%d,%d,%d,%.3f,%.3f' % (name[:80], gx, gy, gz, n, avg / 1e3, mn / 1e3))
else:
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print('"%s",%d,%.3f,%.3f,%.3f' % (name, calls, total, avg, pct))
