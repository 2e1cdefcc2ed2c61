#!/bin/bash
# A/B of the wave priority of the latency kernels, same box, alternating runs: libmorb.so (all tagged kernels at priority 3),
# libmorb_prionarrow.so (only the few-workgroup kernels), libmorb_noprio.so (nobody)
run() { lib=$1; c=$2; MORB_LIB_PATH=$lib timeout 150 python bench.py --config $c --no-roofline --no-cpu --no-dropin 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$(basename $lib) c$c', d['value'], d.get('value_c_abi_loop'), d.get('latency_ms_isolated'))"; }
L=multi_orb_slam_amd/lib
for c in 1 4; do for rep in 1 2 3; do for v in libmorb.so libmorb_prionarrow.so libmorb_noprio.so; do run $L/$v $c; done; done; done
