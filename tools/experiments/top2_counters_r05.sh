#!/bin/bash
# SQ counters of the exhaustive top-2 kernel (M1, FP4 form, Q = R = 32 000): where do a wave's cycles go?
# usage (on the GPU box): bash tools/top2_counters.sh      -> gpurun_out/top2_counters.txt
R=$(cd "$(dirname "$0")/../.." && pwd); O=$R/gpurun_out/top2_counters; rm -rf $O; mkdir -p $O
cat > $O/run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth
n = 32000
mt = m.Matcher(); st = mt.stream
d = synth.descriptors(n, 4242)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
res = [rt.DeviceBuffer(n * 4) for _ in range(3)]
scr = rt.DeviceBuffer(max(m.Matcher.top2_scratch_bytes(n, n), 16))
for _ in range(12):
    m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
rt.stream_sync(st)
PY
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $O/run.py > $O/p$i.out 2>&1 < /dev/null
done
python3 - $O <<'PY' > $R/gpurun_out/top2_counters.txt
import sys, glob, csv, collections, os
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in acc.items():
    if "top2" not in k: continue
    print(k)
    for name in sorted(c):
        print("   %-32s %16.0f per launch   %12.1f per wave" % (name, c[name] / n[k][name], c[name] / n[k][name] / max(acc[k]["SQ_WAVES"] / n[k]["SQ_WAVES"], 1)))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $R/gpurun_out/top2_counters.txt
