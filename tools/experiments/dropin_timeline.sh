#!/bin/bash
# kernel timeline of the drop-in class path (host/test_host dropin: 2 x operator() + SearchByProjection per step): which kernels,
# how long, and the gaps (host time) between them.  dropin_timeline.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd $R && python3 - $O <<'P'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import dropin_leg, subprocess
real = subprocess.check_call
def fake(cmd, *a, **k):            # keep the stream file, run the binary ourselves below
    open(os.path.join(sys.argv[1], "cmd.txt"), "w").write(" ".join(cmd))
    return real(cmd, *a, **k)
subprocess.check_call = fake
r = dropin_leg.run(iters=100, warmup=20, check=False, workdir=sys.argv[1])
print({k: r[k] for k in ("extract_cam0_us", "extract_cam1_us", "search_by_projection_us", "search_breakdown_us", "class_calls_us")})
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_dropin -o dropin -- $(cat $O/cmd.txt) > $O/dropin_rocprof.out 2> $O/dropin_rocprof.err
python3 - $(find $O/prof_dropin -name "*.db" | head -1) <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
k = len(rows) // 2
while "pyramid" not in rows[k][0] and "ingest" not in rows[k][0]: k += 1
t0 = rows[k][1]; prev = None
for name, a, b in rows[k:k + 26]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-36:]
    print("%8.1f %8.1f  dur %6.1f  gap %6.1f  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, 0 if prev is None else (a - prev) / 1e3, short))
    prev = b
P
