cd $GRAFT_REPO_ROOT
timeout 300 python -X faulthandler tools/step_breakdown.py > gpurun_out/dbg.txt 2>&1; grep -v "^  File\|^$" gpurun_out/dbg.txt | head -30 | cut -c1-300
MORB_GRAPH_FORK=1 timeout 300 python -X faulthandler tools/step_breakdown.py > gpurun_out/dbg.txt 2>&1; grep -v "^  File\|^$" gpurun_out/dbg.txt | head -30 | cut -c1-300
MORB_STEP_GRAPH=0 timeout 300 python -X faulthandler tools/step_breakdown.py > gpurun_out/dbg.txt 2>&1; grep -v "^  File\|^$" gpurun_out/dbg.txt | head -30 | cut -c1-300
