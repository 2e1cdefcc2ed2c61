# rocprofv3 passes of round 1: kernel-trace stats of bench.py, then PMC passes (own runs, counters only) of the matcher.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/matcher -o matcher -- python3 tools/profile_matcher.py > $O/matcher.out 2> $O/matcher.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 tools/profile_matcher.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 tools/profile_matcher.py > /dev/null 2> $O/pmc_write.err
find $O -type f | head -40
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -15 $f; done
for f in $(find $O -name "*counter_collection.csv"); do echo "== $f"; head -3 $f; wc -l $f; done
