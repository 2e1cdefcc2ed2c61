cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py --no-cpu --no-roofline 2>&1 | grep metric | cut -c1-200
MORB_FORCE_DIST=1 timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu --no-roofline 2>&1 | grep "metric\|rror" | cut -c1-200
