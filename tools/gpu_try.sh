cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 600 python tools/step_breakdown.py 2>&1 | tail -1 | cut -c1-140
for i in 1 2 3; do python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"; done
