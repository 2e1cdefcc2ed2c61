cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 600 python tools/big_breakdown.py 2>&1 | tail -1 | cut -c1-330
timeout 300 python bench.py --no-cpu --no-roofline 2>&1 | grep metric | cut -c1-200
