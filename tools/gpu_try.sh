cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -4
MORB_LIB_PATH=$GRAFT_REPO_ROOT/multi_orb_slam_amd/lib/libmorb_phases.so timeout 200 python tools/phase_clocks.py 2>&1 | grep -A1 resolve
timeout 300 python bench.py --no-cpu --no-roofline 2>&1 | grep metric | cut -c1-200
