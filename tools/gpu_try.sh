cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -12
