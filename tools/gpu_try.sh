cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2; do MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_phases.so timeout 300 python tools/phase_clocks.py 2>&1 | grep -A3 "^resolve" | grep -v sweep; done
for i in 1 2 3; do python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"; done
