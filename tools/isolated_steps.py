#!/usr/bin/env python3
"""N isolated timesteps (no look-ahead) of configs[1] -- the workload of `latency_ms_isolated`; run under
`rocprofv3 --kernel-trace` to get the kernel timeline of one step (tools/step_timeline.py reads the trace)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H, NF, NC = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (640, 480, 1000, 2)   # e.g. 1920 1080 4000 8 = configs[4]
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
dev = [[rt.DeviceBuffer(W * H) for c in range(NC)] for t in range(8)]
for t in range(8):
    for c in range(NC):
        dev[t][c].upload(synth.image(c, t, W, H))
rt.device_sync()
fe.copy_results = False
ts = []
for it in range(N):
    t0 = time.perf_counter()
    r = fe.step([(dev[it % 8][c].ptr, W) for c in range(NC)], resident=True)
    ts.append(time.perf_counter() - t0)
    time.sleep(0.002)          # steps well apart in the trace
ts = sorted(ts[10:])
print(json.dumps({"isolated_step_us_median": round(1e6 * ts[len(ts) // 2], 1), "host_us": [round(x, 1) for x in r["host_us"]]}))
fe.close()
