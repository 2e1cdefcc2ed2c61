cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed"
