#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --kernel-trace run (the *_kernel_trace.csv under the output directory): calls, average / min / max
duration in us, calls per step.   python tools/kernel_stats_summary.py <dir> [steps]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
acc = collections.OrderedDict()
for path in f:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if name.startswith("void "): name = name[5:]
        name = name.split("(")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
        a = acc.setdefault(name, [0, 0.0, 1e30, 0.0]); a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = 0.0
for name, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if a[0] < steps / 2: continue
    print("%-60s calls %6d  per step %5.2f  avg %8.2f us  min %8.2f  max %8.2f  us/step %8.2f" % (name[:60], a[0], a[0] / steps, a[1] / a[0], a[2], a[3], a[1] / steps))
    tot += a[1] / steps
print("sum of kernel time per step: %.1f us" % tot)
