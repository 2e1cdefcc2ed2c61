cd $GRAFT_REPO_ROOT
for i in 1 2; do MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_phases.so timeout 300 python tools/phase_clocks.py 2>&1 | grep -A3 "^resolve"; done
