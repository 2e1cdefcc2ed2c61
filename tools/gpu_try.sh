cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_frontend.py -m gpu -q -x -k full_size 2>&1 | tail -15
