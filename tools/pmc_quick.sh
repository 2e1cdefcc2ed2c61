#!/bin/bash
# Quick counter check of the extractor kernels for one configuration: FETCH_SIZE + WRITE_SIZE passes (and a kernel trace) over
# tools/profile_extractor.py, printed per kernel and timestep.  usage: bash tools/pmc_quick.sh <config 1|2|4> [steps]
C=${1:-4}; STEPS=${2:-8}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_quick_c$C; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1 MORB_CHAIN_GRAPH=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfg${C}_fetch -o p -- python3 $R/tools/profile_extractor.py $C $STEPS > $O/fetch.out 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cfg${C}_write -o p -- python3 $R/tools/profile_extractor.py $C $STEPS > $O/write.out 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 $R/tools/profile_extractor.py $C 40 > $O/trace.out 2>&1
python3 - $O $C $STEPS <<'PY'
import sys, glob, csv, collections, os
O, C, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
def load(tag, counter):
    f = glob.glob(os.path.join(O, "cfg%s_%s" % (C, tag), "**", "*counter_collection.csv"), recursive=True)[0]
    out = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]] += float(r["Counter_Value"])
    return out
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
st = glob.glob(os.path.join(O, "trace", "**", "*kernel_stats.csv"), recursive=True)
avg = {}
if st:
    for r in csv.DictReader(open(st[0])):
        avg[r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print("config %s, per timestep (FETCH_SIZE KiB x 2 x 1024; WRITE_SIZE KiB x 1024):" % C)
for k in sorted(fe, key=lambda k: -fe[k]):
    if k.startswith("__amd"): continue
    a = avg.get(k, (0, 0))
    print("  %-28s fetch %8.2f MB  write %7.2f MB   avg %7.1f us x %d calls" % (k, fe[k] * 2048 / steps / 1e6, wr.get(k, 0) * 1024 / steps / 1e6, a[0], a[1]))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
