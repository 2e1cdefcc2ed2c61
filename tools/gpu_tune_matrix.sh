cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_matcher.py -m gpu -q -k "matrix or top2" 2>&1 | tail -3
for qpb in 32 64 128 256 512; do echo "QPB=$qpb"; MORB_MATRIX_QPB=$qpb SIZES=32000 python tools/bench_kernels.py; done
SIZES=1000,4000,8000 python tools/bench_kernels.py
