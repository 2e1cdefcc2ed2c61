#!/bin/bash
# configs[4]: the resize chain (MORB_TILED_PYRAMID=0) against the tiled single launch: pyramid kernels' durations and step time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export MORB_TILED_PYRAMID=$v
  rm -rf $O/prof_pyr
  MORB_NO_BAR_STAGING=1 MORB_CHAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pyr -o p -- python3 $R/bench.py --config ${CFG:-4} --no-roofline --no-cpu --steps 40 --min-time 0.05 > $O/pyr.json 2> $O/pyr.err
  python3 - $(find $O/prof_pyr -name "*kernel_stats.csv" | head -1) "$v" $O/pyr.json <<'P'
import csv, sys, json
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("MORB_TILED_PYRAMID=%s step ms (under rocprof) %s" % (sys.argv[2], d["ms_per_step"]))
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("pyramid", "resize", "ingest")):
        print("   ", r["Name"].replace("(anonymous namespace)::", "")[:44], "calls", r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
P
done
