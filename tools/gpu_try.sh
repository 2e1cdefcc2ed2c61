cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_frontend.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
for v in 1 0; do MORB_PYRAMID_TAIL=$v python3 bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail=$v', d['value'], d['ms_per_step'], d['overlap'][-32:], d['extractor_stage_us'])"; done
