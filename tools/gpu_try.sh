cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head
mkdir -p gpurun_out/soak
MORB_POLL=0 timeout 600 python tools/soak_poll.py 30000 gpurun_out/soak/ref.txt 2>&1 | grep digest
for i in 1 2 3; do MORB_POLL=1 timeout 600 python tools/soak_poll.py 30000 gpurun_out/soak/p$i.txt 2>&1 | grep digest; diff gpurun_out/soak/ref.txt gpurun_out/soak/p$i.txt | head -4; done
rm -f gpurun_out/soak/*.txt
for i in 1 2; do python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('poll', d['value'], d['ms_per_step'])"; MORB_POLL=0 python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sync', d['value'], d['ms_per_step'])"; done
