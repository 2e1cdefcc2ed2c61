#!/bin/bash
# SQ counters of k_hamming_top2_mfma (is the matrix pipe busy while the vector ALU works?).  top2_pmc.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU"
rocprofv3 --pmc $P1 --output-format csv -d $O/top2_p1 -o p -- python3 $R/tools/experiments/top2_launch.py > $O/top2_p1.out 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $O/top2_p2 -o p -- python3 $R/tools/experiments/top2_launch.py > $O/top2_p2.out 2>&1
python3 - $O <<'P'
import csv, sys, glob, collections
for d in ("top2_p1", "top2_p2"):
    f = glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "top2_mfma" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, c) in sorted(acc.items()): print("%-32s %16.0f per launch (%d)" % (k, s / c, c))
P
