cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_configs.py 2>&1 | tail -12 | cut -c1-120
