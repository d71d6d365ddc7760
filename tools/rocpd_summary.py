"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table (CSV on stdout)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                  "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("kernel,calls,total_ms,avg_ms,min_ms,max_ms,pct,vgpr,agpr,sgpr,lds_bytes,grid_x,wg_x")
for r in rows:
    print('"%s",%d,%.3f,%.4f,%.4f,%.4f,%.2f,%s,%s,%s,%s,%s,%s' % (r[0][:110], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6,
                                                             100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11]))
