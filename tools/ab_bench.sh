#!/bin/bash
# A/B of bench.py's headline numbers under environment toggles:  tools/ab_bench.sh "VAR=0" "VAR=1" ...   (each argument is one
# run's environment; "-" = none)
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs python bench.py --steps 1000 --warmup 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s value %8.1f  isolated %.4f (p5 %.4f p95 %.4f)  h2d %8.1f  dropin %7.1f' % ('$cfg', d['value'], d['latency_ms_isolated'], d['latency_isolated']['p5'], d['latency_isolated']['p95'], d['value_h2d_inclusive'], d['dropin_fps']))"
done
