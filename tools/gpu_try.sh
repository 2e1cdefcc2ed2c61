cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  (cd build/ab/old && python3 bench.py --steps 3000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'])")
  python3 bench.py --steps 3000 --warmup 200 --no-cpu --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"
done
