#!/usr/bin/env python
"""Summarise a rocprofv3 results.db (kernel trace) into a per-kernel table (stdout, markdown-ish)."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
db = sqlite3.connect(dbs[0])
cur = db.cursor()
rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                   "from kernels group by name order by sum(end-start) desc").fetchall()
tot = sum(r[5] for r in rows)
span = cur.execute("select min(start), max(end) from kernels").fetchone()
print("kernel-time total %.3f ms over a %.3f ms span (GPU busy %.1f%%), %d dispatches" %
      (tot / 1e6, (span[1] - span[0]) / 1e6, 100.0 * tot / (span[1] - span[0]), sum(r[1] for r in rows)))
print("%-100s %8s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share"))
for r in rows[:45]:
    print("%-100s %8d %10.2f %10.2f %10.2f %6.1f%%" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100 * r[5] / tot))
