cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python tools/bench_configs.py 2>&1 | tail -6 | cut -c1-130
