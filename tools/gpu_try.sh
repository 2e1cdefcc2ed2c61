cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python tools/dbg_fallback.py 2>&1 | tail -5 | cut -c1-60 | tr '\n' ';'; echo; done
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -5
