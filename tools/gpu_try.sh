cd $GRAFT_REPO_ROOT
MORB_FORCE_DIST=1 timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu --no-roofline 2>&1 | grep "metric\|Error\|error" | cut -c1-200
