cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 600 python tools/bench_configs.py 2>&1 | tail -2 | cut -c1-200
