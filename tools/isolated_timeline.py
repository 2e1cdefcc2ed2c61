#!/usr/bin/env python3
"""Kernel timeline of ONE isolated timestep from a `rocprofv3 --kernel-trace --output-format csv` run of tools/isolated_steps.py
(steps are 2 ms apart in the trace):   python tools/isolated_timeline.py <..._kernel_trace.csv> [step #]
start and end of every kernel relative to the step's first kernel, per queue."""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
steps, cur = [], []
for r in rows:
    if cur and r[0] - max(x[1] for x in cur) > 1_000_000:
        steps.append(cur); cur = []
    cur.append(r)
steps.append(cur)
st = steps[int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 2]
t0 = st[0][0]
for s, e, n, q in st:
    mm = re.search(r"k_[A-Za-z0-9_]+", n)
    name = mm.group(0) if mm else n.split("(")[0]
    print("%8.1f -> %8.1f  (%6.1f us)  queue %-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:60]))
print("step: %.1f us of kernels end to end" % ((max(x[1] for x in st) - t0) / 1e3))
