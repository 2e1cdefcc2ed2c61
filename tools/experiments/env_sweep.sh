#!/bin/bash
# bench value of configs[1] for several values of one environment variable: env_sweep.sh <outdir> <VAR> v1 v2 ... (each run twice, interleaved)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O; V=$2; shift 2
for rep in 1 2; do for n in "$@"; do
  env $V=$n python3 $R/bench.py --no-dropin --no-roofline --no-cpu 2> $O/sweep.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', '$n', 'value', d['value'], 'ms/step', d['ms_per_step'], 'isolated', d.get('latency_ms_isolated'), 'c_abi_loop', d.get('value_c_abi_loop'))"
done; done
