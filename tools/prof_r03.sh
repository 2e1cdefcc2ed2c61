#!/bin/bash
# round-3 quick profile: kernel stats of configs[4] and configs[1] bench runs + timeline of isolated configs[1] steps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MORB_NO_BAR_STAGING=1
export MORB_CHAIN_GRAPH=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 $R/bench.py --config 4 --no-roofline --no-cpu > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c1 -o c1 -- python3 $R/bench.py --no-roofline --no-cpu --no-dropin > $O/bench_c1_under_rocprof.json 2> $O/prof_c1.err
rocprofv3 --kernel-trace -d $O/prof_iso -o iso -- python3 $R/tools/isolated_steps.py 60 > $O/iso.json 2> $O/prof_iso.err
python3 $R/tools/step_timeline.py $(find $O/prof_iso -name "*.db" | head -1) > $O/iso_timeline.txt 2>&1
find $O -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -25 "$f"; done
cat $O/iso_timeline.txt
