#!/bin/bash
# per-kernel averages of a short configs[1] bench run under overlap (two steps announced ahead).  c1_kernels.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1 MORB_CHAIN_GRAPH=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c1 -o c1 -- python3 $R/bench.py --no-roofline --no-cpu --no-dropin --min-time 0.3 > $O/c1_under_rocprof.json 2> $O/prof_c1.err
python3 - $(find $O/prof_c1 -name "*kernel_stats.csv" | head -1) <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-48s calls %6s avg %9.1f us  min %7.1f  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Percentage"]))
P
