#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes over the EXTRACTOR and step kernels (run under
`rocprofv3 --pmc ... -- python3 tools/profile_extractor.py <config> <steps>`; tools/collect_pmc_extractor.sh drives the passes).

<config> = 1 | 2 | 3 | 4 (BASELINE.json configs[1] 2 x 640x480 @1000, configs[2] 2 x 1280x720 @2000, configs[3] 4 x 640x480 @1000 on one GPU,
configs[4] 8 x 1920x1080 @4000).
Launches a 1 GiB hipMemset and a 1 GiB device copy (calibration of WRITE_SIZE / FETCH_SIZE, MI355X_MICROARCH.md section HBM),
then <steps> ISOLATED front-end timesteps on HBM-resident images (every step: k_ingest, pyramid, k_fast_cells, k_octree,
k_describe, frame grid, k_project[_side], k_top2_merge, k_resolve / k_rs_*).  Counters are summed per kernel name and divided by
<steps> by tools/parse_pmc_extractor.py: bytes / instructions PER TIMESTEP."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth, pipeline

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, NF, NC = {1: (640, 480, 1000, 2), 2: (1280, 720, 2000, 2), 3: (640, 480, 1000, 4), 4: (1920, 1080, 4000, 8)}[cfg]
GIB = 1 << 30
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=NF)] * NC, W, H)
a = rt.DeviceBuffer(GIB); b = rt.DeviceBuffer(GIB)
rt._L().orb_memset(a.ptr, 1, GIB, fe.stream)
rt._L().orb_memcpy_d2d(b.ptr, a.ptr, GIB, fe.stream)
rt.stream_sync(fe.stream)
RING = 4
dev = [[rt.DeviceBuffer(W * H) for c in range(NC)] for t in range(RING)]
for t in range(RING):
    for c in range(NC):
        dev[t][c].upload(synth.image(c, t, W, H))
rt.device_sync()
fe.copy_results = False
n = 0
for t in range(steps):
    r = fe.step([(dev[t % RING][c].ptr, W) for c in range(NC)], resident=True)
    n = r["n_total"]
print("profile_extractor: config %d, %d isolated steps, %d features in the last one" % (cfg, steps, n))
fe.close()
