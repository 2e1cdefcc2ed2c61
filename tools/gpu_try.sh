cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_frontend.py tests/test_gpu_host_cpp.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2; do MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_phases.so timeout 300 python tools/phase_clocks.py 2>&1 | grep -A3 "^resolve"; done
