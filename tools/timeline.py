#!/usr/bin/env python3
"""Two-stream timeline of one train step from a rocprofv3 --kernel-trace database (rocpd SQLite):
   python tools/timeline.py <results.db> [step_index_from_end]
Prints, for the step that ends with the chosen adam_kernel dispatch, when each queue (HIP stream) is busy and what the end of
the step waits for.  Caveat: on this ROCm rocprofv3's kernel trace serialises the dispatches of both streams onto one queue
(the traced step takes 9 ms instead of 4.7), so the trace shows kernel durations and order but NOT the overlap; the overlap is
measured with HIP events by tools/tail_probe.py."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
print("columns:", ', '.join(cols))
rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or '0')).fetchall()
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'^void ', '', re.sub(r'\(anonymous namespace\)::', '', n)))[:48]
adams = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
i1 = adams[-back]
i0 = adams[-back - 1] + 1
step = rows[i0:i1 + 1]
t0 = step[0][1]
print("step: %d dispatches, %.1f us from first start to adam end" % (len(step), (step[-1][2] - t0) / 1e3))
queues = {}
for n, s, e, q in step:
    queues.setdefault(q, []).append((n, s, e))
for q, ks in sorted(queues.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for _, s, e in ks) / 1e3
    print("queue %s: %4d kernels, first start %8.1f us, last end %8.1f us, busy %8.1f us" %
          (q, len(ks), (ks[0][1] - t0) / 1e3, (max(e for _, _, e in ks) - t0) / 1e3, busy))
print("last 14 dispatches of the step (start / end relative to the step start, us):")
for n, s, e, q in step[-14:]:
    print("  q%-3s %9.1f %9.1f  %6.1f  %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short(n)))
# idle gaps of the busiest queue (the main chain): time between a kernel's end and the next start
main_q = max(queues.items(), key=lambda kv: len(kv[1]))[0]
ks = queues[main_q]
gaps = [(ks[i + 1][1] - ks[i][2]) / 1e3 for i in range(len(ks) - 1)]
print("main queue: sum of kernel time %.1f us, sum of gaps %.1f us (median gap %.2f us, %d gaps > 10 us)" %
      (sum(e - s for _, s, e in ks) / 1e3, sum(gaps), sorted(gaps)[len(gaps) // 2], sum(g > 10 for g in gaps)))
big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, i in big:
    print("  gap %7.1f us after %s (ends %.1f us)" % (g, short(ks[i][0]), (ks[i][2] - t0) / 1e3))
