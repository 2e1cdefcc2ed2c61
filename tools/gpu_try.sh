cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu 2>&1 | grep '^{' | tail -1 | cut -c1-200
