cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
MORB_LIB_PATH=$GRAFT_REPO_ROOT/multi_orb_slam_amd/lib/libmorb_phases.so timeout 200 python tools/phase_clocks.py
timeout 300 python tools/step_breakdown.py 2>&1 | tail -1 | cut -c1-500
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-300
