cd $GRAFT_REPO_ROOT
timeout 300 python -X faulthandler -m pytest tests/test_gpu_extractor.py -m gpu -q -x > gpurun_out/dbg_ex.txt 2>&1; grep -E "passed|failed|Error|error|assert|Fatal|Memory" gpurun_out/dbg_ex.txt | head -8
