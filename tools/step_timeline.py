#!/usr/bin/env python3
"""Kernel timeline of one isolated timestep from a rocprofv3 --kernel-trace database (rocpd sqlite): start / end of every
kernel relative to the step's first kernel, the gaps between dependent kernels, per-stream."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cur = db.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall() if "stream_id" in cols else \
    [(r[0], r[1], r[2], 0) for r in db.execute("select name, start, end from kernels order by start")]
# split into steps at gaps > 1 ms
steps, curstep = [], []
for r in rows:
    if curstep and r[1] - curstep[-1][2] > 1_000_000:
        steps.append(curstep); curstep = []
    curstep.append(r)
steps.append(curstep)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 2
st = steps[which]
t0 = st[0][1]
print("steps in trace: %d; step %d: %d kernels, span %.1f us" % (len(steps), which, len(st), (max(r[2] for r in st) - t0) / 1e3))
for name, a, b, sid in st:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
    print("%8.1f %8.1f  dur %6.1f  stream %-4s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, sid, short))
