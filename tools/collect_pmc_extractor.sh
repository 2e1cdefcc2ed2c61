#!/bin/bash
# Counter passes over the extractor / step kernels: FETCH_SIZE, WRITE_SIZE and one SQ pass per configuration,
# each its own rocprofv3 run (--pmc only, no trace domains), the program itself right behind `--`.
# usage (GPU box): bash tools/collect_pmc_extractor.sh <out subdir under gpurun_out/r03> [steps]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; STEPS=${2:-8}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1
export MORB_CHAIN_GRAPH=0
rocprofv3 -L > $O/counters_list.txt 2>&1
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for c in 1 2 4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfg${c}_fetch -o p -- python3 $R/tools/profile_extractor.py $c $STEPS > $O/cfg${c}_fetch.out 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cfg${c}_write -o p -- python3 $R/tools/profile_extractor.py $c $STEPS > $O/cfg${c}_write.out 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $O/cfg${c}_sq -o p -- python3 $R/tools/profile_extractor.py $c $STEPS > $O/cfg${c}_sq.out 2>&1
done
python3 $R/tools/parse_pmc_extractor.py $O $STEPS $O/pmc_traffic.json > $O/parse.out 2>&1
tail -n 5 $O/cfg1_sq.out $O/parse.out
# the raw CSVs are large: keep the parsed JSON and the per-kernel sums only
find $O -name "*counter_collection.csv" -size +8M -delete
