#!/bin/bash
# Kernel trace of the class-API leg (2 x ORBextractor::operator() + ORBmatcher::SearchByProjection per frame): writes
# gpurun_out/dropin_trace/{kernel trace csv, timeline.txt}.  Run on the GPU box from the repo root.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/dropin_trace
rm -rf $O; mkdir -p $O
python3 - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import dropin_leg, subprocess
real = subprocess.check_call
def fake(cmd, **kw):    # keep the stream file, skip the run: the traced run follows
    import shutil
    shutil.copy(cmd[2], "gpurun_out/dropin_trace/stream.bin")
    raise SystemExit(0)
subprocess.check_call = fake
dropin_leg.run(check=False, iters=40, warmup=10)
PY
export MORB_CHAIN_GRAPH=0 MORB_NO_BAR_STAGING=${MORB_NO_BAR_STAGING:-1}
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- multi_orb_slam_amd/host/test_host dropin $O/stream.bin $O/out.bin ${BATCH:-0} > $O/run.log 2>&1 || true
CSV=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$CSV" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
# frames: split at gaps > 60 us of host work between the search and the next extraction is not reliable; print a window
mid = len(ks) * 3 // 4
t0 = ks[mid][1]
for name, a, b, q in ks[mid:mid + 60]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    print("%9.1f %9.1f  dur %6.1f  q %-6s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, short))
PY
rm -f $O/stream.bin $O/out.bin
cat $O/timeline.txt | head -70
