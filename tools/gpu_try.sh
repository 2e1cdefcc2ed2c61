cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -8
