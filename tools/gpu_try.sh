cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -5
MORB_LIB_PATH=$GRAFT_REPO_ROOT/multi_orb_slam_amd/lib/libmorb_phases.so timeout 200 python tools/phase_clocks.py 2>&1 | grep resolve
timeout 300 python bench.py --no-cpu --no-roofline 2>&1 | grep metric | cut -c1-200
timeout 300 python bench.py --no-cpu --no-roofline --no-overlap --steps 500 2>&1 | grep metric | cut -c1-200
