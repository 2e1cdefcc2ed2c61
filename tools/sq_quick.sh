#!/bin/bash
# SQ instruction counters per kernel and wave for one configuration (tools/profile_extractor.py under rocprofv3 --pmc).
# usage: bash tools/sq_quick.sh <config 1|2|4> [kernel name filter]
C=${1:-4}; FILT=${2:-k_}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/sq_quick_c$C; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1 MORB_CHAIN_GRAPH=0
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq -o p -- python3 $R/tools/profile_extractor.py $C 8 > $O/sq.out 2>&1
python3 - $O "$FILT" <<'PY'
import sys, glob, csv, collections, os
O, filt = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(O, "sq", "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if filt not in k or not c.get("SQ_WAVES"): continue
    w = c["SQ_WAVES"]
    print("  %-26s waves/step %8.0f  per wave: VALU %7.1f SALU %6.1f LDS %6.1f VMEM_RD %5.1f VMEM_WR %5.1f  wave-cycles %8.0f" % (
        k, w / 8, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w, c["SQ_INSTS_VMEM_WR"] / w, c["SQ_WAVE_CYCLES"] / w))
PY
find $O -name "*counter_collection.csv" -delete
