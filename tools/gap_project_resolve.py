#!/usr/bin/env python3
"""The boundary between the projection and the resolve of a step's search, from a rocprofv3 kernel trace (CSV):
   python tools/gap_project_resolve.py <..._kernel_trace.csv>
prints the distribution of (start of k_resolve_* - end of the k_project* before it on the same queue) and both kernels' durations."""
import csv
import sys
import numpy as np


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "k_project" in n or "k_resolve" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "P" if "k_project" in n else "R", r.get("Queue_Id", "")))
    rows.sort()
    gaps, dp, dr = [], [], []
    last = None
    for s, e, k, q in rows:
        if k == "P":
            last = (s, e, q)
        elif last is not None:
            gaps.append(s - last[1]); dp.append(last[1] - last[0]); dr.append(e - s)
            last = None
    for name, v in (("gap project end -> resolve start", gaps), ("k_project", dp), ("k_resolve", dr)):
        a = np.array(v, dtype=np.float64) / 1e3
        print("%-34s n=%d  p5 %.2f  median %.2f  p95 %.2f us" % (name, len(a), *np.percentile(a, [5, 50, 95])))


if __name__ == "__main__":
    main(sys.argv[1])
