"""Per-kernel totals from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`).
usage: python tools/rocpd_summary.py results.db [steps=1] [top=40] [skip_substring ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
skip = sys.argv[4:]
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
rows = [r for r in rows if not any(s in r[0] for s in skip)]
print("sum of kernel durations: %.3f ms per step" % (sum(r[2] for r in rows) / steps / 1e6))
for r in rows[:top]:
    print("%-96s %6.1f/step %8.3f ms/step  avg %8.1f us" % (r[0][:96], r[1] / steps, r[2] / steps / 1e6, r[3] / 1e3))
