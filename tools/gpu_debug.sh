cd $GRAFT_REPO_ROOT
timeout 300 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/dbg_all.txt 2>&1; tail -3 gpurun_out/dbg_all.txt
python tools/step_breakdown.py 2>&1 | tail -2
python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu 2>/dev/null | cut -c1-200
