cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -5
timeout 300 python tools/step_breakdown.py 2>&1 | tail -1 | cut -c1-400
MORB_CHAIN_GRAPH=0 timeout 300 python tools/step_breakdown.py 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-300
