#!/bin/bash
# bench A/B of one environment switch: bench_ab.sh <outdir> <ENVVAR> [config...] -- value / isolated latency with ENVVAR unset and =0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; mkdir -p $O; V=$2; shift 2
for c in ${@:-1}; do for v in default 0; do
  if [ $v = 0 ]; then export $V=0; else unset $V; fi
  python3 $R/bench.py --config $c --no-dropin --no-roofline --no-cpu > $O/ab_c${c}_$v.json 2> $O/ab_c${c}_$v.err
  python3 - $O/ab_c${c}_$v.json "$V=$v config $c" <<'P'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "isolated", d.get("latency_ms_isolated"), "c_abi_loop", d.get("value_c_abi_loop"))
P
done; done
