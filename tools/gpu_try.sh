cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  " | head
timeout 300 python tools/step_breakdown.py 2>&1 | tail -1 | cut -c1-170
timeout 300 python bench.py --no-cpu --no-roofline 2>&1 | grep "metric\|rror" | cut -c1-200
timeout 300 python bench.py --no-cpu --no-roofline 2>&1 | grep "metric\|rror" | cut -c1-200
MORB_FORCE_DIST=1 timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>&1 | grep "metric\|rror" | cut -c1-200
