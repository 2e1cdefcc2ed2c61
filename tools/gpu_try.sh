cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --no-roofline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['cpu_baseline'])"
