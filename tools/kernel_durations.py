import csv, sys, numpy as np
from collections import defaultdict
d = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        for key in ("k_project_resolve", "k_project(", "k_project_side", "k_resolve_mono"):
            if key in n:
                d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                break
for k, v in d.items():
    a = np.array(v)
    print("%-20s n=%d p5 %.2f median %.2f p95 %.2f us" % (k, len(a), *np.percentile(a, [5, 50, 95])))
