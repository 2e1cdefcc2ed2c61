#!/bin/bash
# Collects the round's measurements on the GPU box: GPU tests, bench.py for every BASELINE.json configuration, and the
# rocprofv3 summaries that go under profiles/ (kernel trace + stats of the same bench command; separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE as MI355X_MICROARCH.md prescribes).  Usage: gpurun -- 'bash tools/collect_profiles.sh'
# rocprofv3 on this image dies with SIGSEGV inside its interception layer (below orbx_run_impl's HIP calls) when the extractor's
# launch chain is captured / replayed as a graph in a long bench run; the profiled runs therefore issue the chain as plain
# launches (MORB_CHAIN_GRAPH=0) and keep the host-written staging in mapped pinned memory (MORB_NO_BAR_STAGING=1).  The kernels
# and their durations are the same; only the host's launch cost differs, and `value` is never taken from a profiled run.
set -x
ROUND=${ROUND:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND/final; mkdir -p $O
cd $R
if [ "$1" != "prof-only" ]; then
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" > $O/pytest.txt
python bench.py > $O/bench.json 2> $O/bench.err
for c in 2 3 4; do python bench.py --config $c --no-roofline > $O/bench_c$c.json 2> $O/bench_c$c.err; done
# the per-rank step of a multi-GPU job, forced on this one GPU (world-1 RCCL group: one all-gather + rig-wide top-2 per step): at the
# tail of the step's extraction chain (the default, with and without the issuer thread) and behind the step's search (rounds 2-4)
MORB_FORCE_DIST=1 python bench.py --no-dropin --no-roofline --no-cpu > $O/bench_forced_exchange.json 2> $O/bench_forced_exchange.err
# the N > 1 job rehearsed as rank PROCESSES on this one GPU (peer transport: direct writes into IPC-mapped arenas, gloo control plane)
python bench.py --gpus 2 --no-dropin --no-roofline --no-cpu > $O/bench_2ranks_1gpu_config1_peer.json 2> $O/bench_2ranks_1gpu_config1_peer.err
python bench.py --gpus 2 --config 3 --no-roofline --no-cpu > $O/bench_2ranks_1gpu_config3_peer.json 2> $O/bench_2ranks_1gpu_config3_peer.err
python bench.py --gpus 4 --config 3 --no-roofline --no-cpu > $O/bench_4ranks_1gpu_config3_peer.json 2> $O/bench_4ranks_1gpu_config3_peer.err
MORB_FORCE_DIST=1 MORB_EXCHANGE_PLACEMENT=inline python bench.py --no-dropin --no-roofline --no-cpu > $O/bench_forced_exchange_inline.json 2> $O/bench_forced_exchange_inline.err
fi
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1
export MORB_CHAIN_GRAPH=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --no-dropin --no-live-traffic > $O/bench_under_rocprof.json 2> $O/prof_bench.err
if [ "$1" != "prof-only" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 $R/bench.py --config 4 --no-roofline --no-cpu > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_matcher -o matcher -- python3 $R/tools/profile_matcher.py > $O/prof_matcher.out 2> $O/prof_matcher.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc_fetch -- python3 $R/tools/profile_matcher.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc_write -- python3 $R/tools/profile_matcher.py > /dev/null 2> $O/pmc_write.err
fi
# matcher traffic (k_hamming_*, k_cross_top2*, k_project, k_resolve, BoW) -> pmc_traffic.json, then the extractor / step kernels
# (FETCH_SIZE, WRITE_SIZE and one SQ pass per configuration) merged into the same file
python3 $R/tools/parse_pmc.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/parse_pmc.out 2>&1
STEPS=8
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for c in 1 2 3 4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfg${c}_fetch -o p -- python3 $R/tools/profile_extractor.py $c $STEPS > $O/cfg${c}_fetch.out 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cfg${c}_write -o p -- python3 $R/tools/profile_extractor.py $c $STEPS > $O/cfg${c}_write.out 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $O/cfg${c}_sq -o p -- python3 $R/tools/profile_extractor.py $c $STEPS > $O/cfg${c}_sq.out 2>&1
done
python3 $R/tools/parse_pmc_extractor.py $O $STEPS $O/pmc_traffic.json > $O/parse_pmc_extractor.out 2>&1
find $O -name "*counter_collection.csv" -size +6M -delete
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*stats.csv"
