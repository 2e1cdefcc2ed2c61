#!/usr/bin/env python3
"""Steady-state timeline out of a rocprofv3 kernel trace csv: per stream, the kernels of one period with start offsets and durations."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*$", "", n)
    r["n"] = n[:40]
rows.sort(key=lambda r: r["s"])
# steps are delimited by k_rs_write (one per search)
marks = [r for r in rows if r["n"].startswith("k_rs_write") or r["n"].startswith("k_resolve")]
print("searches:", len(marks))
if len(marks) > 30:
    per = [(marks[i + 1]["e"] - marks[i]["e"]) / 1000 for i in range(20, len(marks) - 1)]
    per.sort(); print("period us median", per[len(per) // 2], "min", per[0], "max", per[-1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0, t1 = marks[k]["e"], marks[k + 1]["e"]
print("window", (t1 - t0) / 1000, "us")
bystream = collections.defaultdict(list)
for r in rows:
    if r["e"] > t0 - 400000 and r["s"] < t1: bystream[r["Stream_Id"]].append(r)
for sid, rs in sorted(bystream.items()):
    print("== stream", sid, "queue", rs[0]["Queue_Id"])
    for r in rs:
        print("  %9.1f %8.1f  %-40s grid %s wg %s lds %s" % ((r["s"] - t0) / 1000, (r["e"] - r["s"]) / 1000, r["n"], r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"]))
