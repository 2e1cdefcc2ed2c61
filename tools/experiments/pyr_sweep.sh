#!/bin/bash
# configs[4] (8 x 1080p): pyramid launch duration and step time for tile size / workgroup size combinations of k_pyramid_tiled
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for combo in "0 0" "32 256" "64 256" "64 1024" "32 1024"; do
  set -- $combo
  export MORB_PYR_TILE=$1 MORB_PYR_THREADS=$2
  rm -rf $O/prof_pyr
  MORB_NO_BAR_STAGING=1 MORB_CHAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pyr -o p -- python3 $R/bench.py --config 4 --no-roofline --no-cpu --steps 40 --min-time 0.05 > $O/pyr.json 2> $O/pyr.err
  python3 - $(find $O/prof_pyr -name "*kernel_stats.csv" | head -1) "$combo" $O/pyr.json <<'P'
import csv, sys, json
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
for r in csv.DictReader(open(sys.argv[1])):
    if "pyramid" in r["Name"]:
        print("tile/threads", sys.argv[2], ":", r["Name"][:40], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "| step ms (under rocprof)", d["ms_per_step"])
P
done
