cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_host_cpp.py -x -q -m gpu > gpurun_out/pytest_host.log 2>&1
tail -30 gpurun_out/pytest_host.log
