#!/usr/bin/env python3
"""One steady-state period of the overlapped loop from a `rocprofv3 --kernel-trace --output-format csv` run of tools/overlap_host_view.py
(or bench.py): everything between two consecutive k_project starts, per queue, relative to the first.
   python tools/overlap_timeline.py <..._kernel_trace.csv> [period #]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        mm = re.search(r"k_[A-Za-z0-9_]+", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), mm.group(0) if mm else r["Kernel_Name"][:30], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] in ("k_project", "k_project_side")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[k], starts[k + 1]
t0 = rows[a][0]
# kernels that overlap the period [t0, next project start)
t1 = rows[b][0]
for s, e, n, q in rows:
    if e < t0 or s >= t1:
        continue
    print("%8.1f -> %8.1f  (%6.1f us)  queue %-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
print("period: %.1f us" % ((t1 - t0) / 1e3))
