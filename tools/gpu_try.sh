cd $GRAFT_REPO_ROOT
for M in "" ""; do DB_MODE=$M timeout 600 python tools/dist_breakdown.py 2>/dev/null | grep -E "us per step|^  " | tr '\n' ';'; echo; done
