cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_frontend.py tests/test_gpu_host_cpp.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 300 --warmup 30 --no-cpu --no-roofline > $O/bench.json 2> $O/bench.err
grep -E "k_project|k_resolve" $O/bench/bench_kernel_stats.csv | sed -E 's/^"[^"]*(k_[a-z_0-9]+)[^"]*"/\1/' | cut -c1-90
for i in 1 2 3; do python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"; done
