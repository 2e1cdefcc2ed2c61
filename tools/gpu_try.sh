cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_matcher.py tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | tail -2
for t in 2 4 8 12; do echo "threads=$t"; MORB_OCTREE_THREADS=$t python tools/step_breakdown.py 2>&1 | tail -1 | cut -c1-330; done
