#!/usr/bin/env python3
"""The dispatches of ONE train step of a rocprofv3 --kernel-trace database in start order -- duration, gap to the end of the previous
dispatch, stream -- so that two engine plans can be compared launch by launch (tools/step_profile.py under DPP_NO_SIDE_STREAM=1 or not):
   python tools/prof_sequence.py x_results.db [which step, counted from the last = 1] > sequence.txt
A step is delimited by adam_kernel (the one launch every step ends with)."""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import pretty  # noqa: E402

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
stream = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else '0')
rows = db.execute("select name, start, end, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, %s from kernels order by start" % stream).fetchall()
ends = [i for i, r in enumerate(rows) if re.search(r'\badam_kernel\b', r[0])]
if len(ends) < back + 1:
    raise SystemExit("the trace holds %d adam_kernel launches: cannot take step -%d" % (len(ends), back))
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
seq = rows[lo:hi]
t0 = seq[0][1]
print("# %d dispatches, %.1f us from the first start to the last end, %.1f us of kernel time" %
      (len(seq), (max(r[2] for r in seq) - t0) / 1e3, sum(r[2] - r[1] for r in seq) / 1e3))
print("%9s %8s %8s %6s  %s" % ('start_us', 'dur_us', 'gap_us', 'stream', 'kernel <grid>'))
prev_end = {}
for name, s, e, gx, gy, gz, q in seq:
    name = pretty(name)
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print("%9.2f %8.2f %8.2f %6s  %s <%d,%d,%d>" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, name[:90], gx, gy, gz))
