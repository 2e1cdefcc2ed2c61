#!/bin/bash
# (on the GPU box) the defect's A/B matrix: library variant x hardware-queue limit, SEC seconds each; one JSON line per cell in
# gpurun_out/describe_defect.jsonl.  Usage: bash tools/describe_defect/ab.sh SEC cell [cell ...] with cell = variant[:queues[:run_rig options, '+' for spaces]]
SEC=${1:-120}; shift
mkdir -p gpurun_out/r05
for cell in "$@"; do
  IFS=: read -r v q opts <<< "$cell"; opts=${opts//+/ }
  libp=multi_orb_slam_amd/lib/libmorb_$v.so; [ "$v" = product ] && libp=multi_orb_slam_amd/lib/libmorb.so
  echo "== $cell"
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  MORB_LIB_PATH=$PWD/$libp timeout $((SEC + 90)) python3 tools/describe_defect/run_rig.py --seconds $SEC --tag "$cell" $opts --out gpurun_out/describe_defect.jsonl 2>&1 | tail -3 | cut -c1-1500
done
