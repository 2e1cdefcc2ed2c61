cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bow.py tests/test_gpu_host_cpp.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/bow_bench.py 2>&1 | grep -v transform | tail -13 | cut -c1-230
