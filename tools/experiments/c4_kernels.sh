#!/bin/bash
# per-kernel averages of a short configs[4] bench run (8 cameras 1920x1080, 4000 features each).  c4_kernels.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1 MORB_CHAIN_GRAPH=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 $R/bench.py --config ${CFG:-4} --no-roofline --no-cpu --steps 60 --min-time 0.05 > $O/c4_under_rocprof.json 2> $O/prof_c4.err
python3 - $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-48s calls %6s avg %9.1f us  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
