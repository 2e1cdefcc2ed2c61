set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/step_breakdown.py
python bench.py --steps 200 --warmup 20 --no-roofline > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -3 gpurun_out/bench3.err; cat gpurun_out/bench3.json
