cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_matcher.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
for i in 1 2; do MORB_FORCE_DIST=1 python bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist', d['value'], d['ms_per_step'])"; done
python3 bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('single', d['value'], d['ms_per_step'])"
