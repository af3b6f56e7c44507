"""Per-kernel summary of a rocprofv3 rocpd database (gpurun_out/.../*_results.db): python tools/prof_db.py DB [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = db.cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), avg(grid_x*grid_y/workgroup_x), max(vgpr_count), max(lds_size) "
                      "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("total kernel time %.1f ms (%.1f ms / step over %d steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
print("name,calls,total_ms,pct,avg_us,avg_workgroups,vgprs,lds_bytes")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:60]
    print("%s,%d,%.2f,%.1f,%.1f,%.0f,%d,%d" % (n, r[1], r[2] / 1e6, 100 * r[2] / tot, r[3] / 1e3, r[4], r[5], r[6]))
