cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
python tools/step_breakdown.py 2>&1 | tail -1
python bench.py --steps 300 --warmup 30 --no-roofline > gpurun_out/bench4.json 2> gpurun_out/bench4.err; tail -2 gpurun_out/bench4.err; cut -c1-260 gpurun_out/bench4.json
MORB_FORCE_DIST=1 python bench.py --steps 50 --warmup 5 --no-cpu --no-roofline 2>/dev/null | cut -c1-200
