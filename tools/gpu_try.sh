cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bow.py -x -q -m gpu > gpurun_out/pytest_bow.log 2>&1
tail -30 gpurun_out/pytest_bow.log
