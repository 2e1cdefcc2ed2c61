set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 200 --warmup 20 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; tail -3 gpurun_out/bench2.err; cat gpurun_out/bench2.json
MORB_FORCE_DIST=1 python bench.py --steps 50 --warmup 5 --no-cpu --no-roofline > gpurun_out/bench2_dist.json 2> gpurun_out/bench2_dist.err; tail -3 gpurun_out/bench2_dist.err; cat gpurun_out/bench2_dist.json
