#!/bin/bash
# SQ counters of the exhaustive top-2 kernel (separate passes).  usage: bash tools/experiments/top2_counters.sh [lib.so]
R=$(cd "$(dirname "$0")/../.." && pwd); O=$R/gpurun_out/top2_ctr; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export MORB_LIB_PATH=$1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/experiments/top2_once.py 32000 fp4 3 > $O/p$i.out 2>&1 || tail -3 $O/p$i.out
done
python3 - $O <<'PY'
import sys, glob, csv, collections, os
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen: cnt[(k, f)] += 1
for k, c in acc.items():
    if "top2" not in k: continue
    print(k)
    for name, v in sorted(c.items()): print("   %-32s %16.0f  (sum over 3 launches)" % (name, v))
PY
find $O -name "*counter_collection.csv" -delete
