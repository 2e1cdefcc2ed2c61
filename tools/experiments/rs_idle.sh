#!/bin/bash
# (the MORB_RS_MONO / MORB_RS_IDLE switches this script drives were removed from search.hip with the experiment: profiles/r03/notes_experiments.md)
# configs[4]: launches with changes (orbm_debug_last_resolve()[2] - 1) and bench value for several MORB_RS_IDLE settings of k_rs_mono
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
for n in ${IDLES:-0 2 4 8 16 32}; do
  echo "MORB_RS_IDLE=$n: $(MORB_RS_IDLE=$n python3 $R/tools/experiments/c4_sweeps.py 2>&1 | tail -3 | tr '\n' ' ')"
  MORB_RS_IDLE=$n python3 $R/bench.py --config 4 --no-dropin --no-roofline --no-cpu 2> $O/idle.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   value', d['value'], 'ms/step', d['ms_per_step'], 'isolated', d.get('latency_ms_isolated'))"
done
