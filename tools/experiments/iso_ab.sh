#!/bin/bash
# isolated-step timeline A/B: MORB_RESOLVE_MONO=1 (default) vs 0 -- kernel durations and gaps of configs[1] steps run one at a time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export MORB_RESOLVE_MONO=$v
  rocprofv3 --kernel-trace -d $O/prof_iso$v -o iso -- python3 $R/tools/isolated_steps.py 60 > $O/iso$v.json 2> $O/prof_iso$v.err
  python3 $R/tools/step_timeline.py $(find $O/prof_iso$v -name "*.db" | head -1) > $O/iso_timeline$v.txt 2>&1
  echo "== MONO=$v"; cat $O/iso_timeline$v.txt; cat $O/iso$v.json | tail -2
done
