cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/big_breakdown.py 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --no-cpu 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
