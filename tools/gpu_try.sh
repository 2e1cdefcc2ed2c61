cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_matcher.py -m gpu -q -x -k gathered 2>&1 | tail -2
MORB_FORCE_DIST=1 timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>&1 | grep "metric\|rror" | cut -c1-200
MORB_FORCE_DIST=1 timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu --no-roofline 2>&1 | grep "metric\|rror" | cut -c1-200
