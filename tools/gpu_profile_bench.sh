cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu --no-roofline > $O/bench.json 2> $O/bench.err
cat $O/bench/bench_kernel_stats.csv | cut -c1-220
