cd $GRAFT_REPO_ROOT
for i in 1 2; do MORB_FORCE_DIST=1 python bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist', d['value'], d['ms_per_step'])"; done
