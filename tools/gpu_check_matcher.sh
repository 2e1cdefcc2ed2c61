cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_matcher.py -m gpu -q 2>&1 | tail -3
SIZES=4000,31000,32000,32768 python tools/bench_kernels.py
