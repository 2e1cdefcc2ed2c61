cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MORB_EXCHANGE_TIMEOUT_MS=10000
fail=0
for i in $(seq 1 15); do
  port=$((29700 + i))
  if [ $((i % 3)) -eq 0 ]; then export MORB_OCT_MAX_KEYS=4096; args="3 3 640 480 500 12 0 0 aheads=0,1,3 noise_at=2:1"; W=3
  elif [ $((i % 3)) -eq 1 ]; then unset MORB_OCT_MAX_KEYS; args="2 2 640 480 1000 16 3 0"; W=2
  else export MORB_OCT_MAX_KEYS=16384; args="4 4 320 240 300 16 0 0 aheads=3,0,2,1 noise_at=5:2 slow=2:5"; W=4; fi
  set -- $args
  for r in $(seq 0 $((W-1))); do python tests/mp_rank.py $1 $r $port ${@:2} > /tmp/mp_$r.txt 2>&1 & done
  wait
  if grep -q "bit-exact vs the oracle" /tmp/mp_0.txt; then echo "run $i ($args): ok $(grep -o 'median step.*' /tmp/mp_0.txt)"; else echo "run $i ($args): FAILED"; fail=1; for r in $(seq 0 $((W-1))); do grep -v "amdgpu.ids\|socket.cpp\|Gloo" /tmp/mp_$r.txt | tail -3; done; fi
done
echo fail=$fail
