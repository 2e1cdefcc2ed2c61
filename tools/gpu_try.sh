cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bow.py tests/test_gpu_host_cpp.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
