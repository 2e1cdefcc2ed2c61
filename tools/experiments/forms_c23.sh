#!/bin/bash
# configs[2] / [3]: which pyramid and FAST forms are best for mid-size rigs?
run() { c=$1; shift; env "$@" timeout 120 python bench.py --config $c --no-roofline --no-cpu 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c$c $*', d['value'], d.get('latency_ms_isolated'), d.get('extractor_stage_us'))"; }
for c in 2 3; do
run $c X=1
run $c MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0
run $c MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_T4_W=64 MORB_PYR_T4_H=32
run $c MORB_PYR_CHAIN=1 MORB_PYRAMID_PAIRS=0 MORB_PYR_SPLIT=99 MORB_PYR_T4_W=64 MORB_PYR_T4_H=32
run $c MORB_FAST_FORM=1
run $c MORB_FAST_FORM=2
done
