#!/usr/bin/env python
"""Prints a per-queue timeline of a few steady-state agent steps from a rocprofv3 results.db."""
import glob
import sqlite3
import sys

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200   # rmsprop launches to skip (warm-up)
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dbs = glob.glob(path + "/**/*_results.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select name, start, end, queue_id, stream_id, grid_x, grid_y, grid_z from kernels order by start").fetchall()
marker = sys.argv[4].split("|") if len(sys.argv) > 4 else ["rmsprop_step", "late_step_kernel", "clip_step_kernel"]   # the update's last kernel
steps = [i for i, r in enumerate(rows) if any(m in r[0] for m in marker)]
i0, i1 = steps[skip] + 1, steps[skip + nsteps] + 1
t0 = rows[i0][1]
print("window: %d kernels, %.1f us" % (i1 - i0, (rows[i1 - 1][2] - t0) / 1e3))
prev_end = {}
for r in rows[i0:i1]:
    name = r[0].split("(")[0].replace("void ", "")[:58]
    q = r[3]
    gap = (r[1] - prev_end.get(q, r[1])) / 1e3
    prev_end[q] = r[2]
    print("q%-3s s%-3s %8.1f -> %8.1f (%6.1f us, gap %6.1f)  grid %5dx%dx%d  %s" %
          (q, r[4], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[5], r[6], r[7], name))
