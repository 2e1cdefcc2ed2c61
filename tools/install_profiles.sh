#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/<round>/final into the tracked profiles/<round>/ (summaries only)
# and the merged counter traffic into profiles/pmc_traffic.json.   usage: bash tools/install_profiles.sh r03
R=${1:-r06}; F=gpurun_out/$R/final; P=profiles/$R; mkdir -p $P
cp $F/bench.json $P/bench.json
for c in 2 3 4; do cp $F/bench_c$c.json $P/bench_config$c.json; done
cp $F/bench_forced_exchange.json $F/bench_forced_exchange_inline.json $F/bench_*ranks_1gpu_*_peer.json $P/ 2>/dev/null
cp $F/bench_under_rocprof.json $P/bench_under_rocprof.json
cp $F/bench_c4_under_rocprof.json $P/bench_config4_under_rocprof.json 2>/dev/null
cp $F/prof_bench/bench_kernel_stats.csv $P/bench_kernel_stats.csv
cp $F/prof_c4/c4_kernel_stats.csv $P/bench_config4_kernel_stats.csv
cp $F/prof_matcher/matcher_kernel_stats.csv $P/matcher_kernel_stats.csv
cp $F/pmc_fetch/pmc_fetch_counter_collection.csv $P/pmc_fetch_counter_collection.csv
cp $F/pmc_write/pmc_write_counter_collection.csv $P/pmc_write_counter_collection.csv
for c in 1 2 3 4; do for k in fetch write; do cp $F/cfg${c}_$k/p_counter_collection.csv $P/pmc_extractor_cfg${c}_${k}_counter_collection.csv; done; done
cp $F/pytest.txt $P/pytest_gpu.txt
cp $F/pmc_traffic.json profiles/pmc_traffic.json
ls -la $P
