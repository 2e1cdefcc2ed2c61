#!/bin/bash
# k_pyramid_tiled4 instead of k_pyramid_tiled on configs[1] (2 x 640x480)?
run() { env "$@" timeout 120 python bench.py --no-roofline --no-cpu 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d.get('value_c_abi_loop'), d.get('latency_ms_isolated'), d.get('extractor_stage_us'))"; }
run X=1
run MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=99 MORB_PYR_T4_W=64 MORB_PYR_T4_H=64
run MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=99 MORB_PYR_T4_W=64 MORB_PYR_T4_H=32
run MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=99 MORB_PYR_T4_W=32 MORB_PYR_T4_H=32
run MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=99 MORB_PYR_T4_W=128 MORB_PYR_T4_H=32
run MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=99 MORB_PYR_T4_W=64 MORB_PYR_T4_H=64 MORB_PYR_T4_NT0=512
run MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=3 MORB_PYR_T4_W=64 MORB_PYR_T4_H=32
