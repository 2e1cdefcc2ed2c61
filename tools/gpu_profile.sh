# rocprofv3 passes: kernel-trace stats of bench.py, kernel stats + PMC passes (own runs, counters only) of the matcher.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
python3 bench.py --steps 300 --warmup 30 > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/matcher -o matcher -- python3 tools/profile_matcher.py > $O/matcher.out 2> $O/matcher.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 tools/profile_matcher.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 tools/profile_matcher.py > /dev/null 2> $O/pmc_write.err
cat $O/bench_plain.json | cut -c1-1500
