#!/bin/bash
# sweep of k_pyramid_tiled4's second-launch tile / workgroup size at configs[4]
run() { env "$@" timeout 120 python bench.py --config 4 --no-roofline --no-cpu 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d.get('latency_ms_isolated'), d.get('extractor_stage_us')['pyramid'])"; }
MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 timeout 100 python tools/experiments/t4_quick.py 2>&1 | grep -c "levels differing: 0"
run X=1
run MORB_PYR_T4_NT1=512
run MORB_PYR_T4_W1=64 MORB_PYR_T4_H1=64
run MORB_PYR_T4_W1=64 MORB_PYR_T4_H1=32
run MORB_PYR_T4_W1=128 MORB_PYR_T4_H1=32
run MORB_PYR_T4_NT0=512 MORB_PYR_T4_NT1=512
run MORB_PYR_T4_W1=64 MORB_PYR_T4_H1=64 MORB_PYR_T4_NT1=512
run MORB_PYR_T4_W=64 MORB_PYR_T4_H=64
run MORB_PYR_T4_W=128 MORB_PYR_T4_H=32
run MORB_PYR_T4_W=256 MORB_PYR_T4_H=32 MORB_PYR_T4_NT0=512
