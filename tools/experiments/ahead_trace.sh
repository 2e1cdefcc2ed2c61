#!/bin/bash
# kernel timeline of the overlapped loop at a look-ahead depth ($2, default 3): which kernels run when, per stream / queue
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O; A=${2:-3}
cd /tmp && export TMPDIR=/tmp
export MORB_CHAIN_GRAPH=${GRAPH:-0}
python3 $R/tools/experiments/overlap_host_us.py $A 2>&1 | tail -1
rm -rf $O/prof_ahead
rocprofv3 --kernel-trace --output-format csv -d $O/prof_ahead -o a -- python3 $R/tools/experiments/overlap_host_us.py $A > $O/ahead.out 2>&1
tail -1 $O/ahead.out
python3 - $(find $O/prof_ahead -name "*kernel_trace.csv" | head -1) <<'P'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:24], r["Queue_Id"], r["Stream_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
k = len(rows) // 2
while "k_project" not in rows[k][2]: k += 1
t0 = rows[k][0]
for a, b, n, q, s in rows[k - 12:k + 40]:
    print("%8.1f %8.1f dur %6.1f  queue %-3s stream %-3s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, s, n))
P
