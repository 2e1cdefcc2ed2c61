#!/bin/bash
# (the MORB_EXTRACT_CU_RESERVE switch this script drives was removed from extractor.hip with the experiment: profiles/r03/notes_experiments.md)
# overlapped throughput / isolated latency of configs[1] and configs[4] with n compute units kept free of extraction kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
for n in 0 16 32 64 0; do for c in 1 4; do
  MORB_EXTRACT_CU_RESERVE=$n python3 $R/bench.py --config $c --no-dropin --no-roofline --no-cpu 2> $O/cu_$n_$c.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('reserve', $n, 'config', $c, 'value', d['value'], 'ms/step', d['ms_per_step'], 'isolated', d.get('latency_ms_isolated'), 'c_abi_loop', d.get('value_c_abi_loop'))"
done; done
