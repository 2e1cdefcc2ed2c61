cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
for i in 1 2 3; do python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('poll', d['value'], d['ms_per_step'])"; MORB_POLL=0 python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sync', d['value'], d['ms_per_step'])"; done
