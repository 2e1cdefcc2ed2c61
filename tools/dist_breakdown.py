#!/usr/bin/env python3
"""bench.py's forced-exchange loop with per-piece host timers (one GPU, world-1 RCCL group)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import torch, torch.distributed as dist
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.dist import DescriptorExchange
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if os.environ.get("DEVID") == "1" else {}))
torch.cuda.set_device(0)
W, H, RING = 640, 480, 8
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
fe.gather = DescriptorExchange(torch.device("cuda", 0), dist); fe.world = 2
host = [[synth.image(c, t, W, H) for c in range(2)] for t in range(RING)]
dev = []
for t in range(RING):
    row = []
    for c in range(2):
        b = rt.DeviceBuffer(W * H); b.upload(host[t][c]); row.append(b)
    dev.append(row)
rt.device_sync()
RES = os.environ.get("RESIDENT", "1") == "1"
arg = (lambda t: [(dev[t % RING][c].ptr, W) for c in range(2)]) if RES else (lambda t: host[t % RING])
fe.copy_results = False
T = {"step": 0.0, "gather": 0.0}
orig_step, orig_gather = fe.fe.step, fe.gather.__call__
def tstep(*a, **k):
    t0 = time.perf_counter(); r = orig_step(*a, **k); T["step"] += time.perf_counter() - t0; return r
class G:
    def __call__(self, f):
        t0 = time.perf_counter(); r = orig_gather(f); T["gather"] += time.perf_counter() - t0; return r
fe.fe.step = tstep; fe.gather = G()
gc.collect(); gc.freeze(); gc.disable()
for i in range(50):
    fe.step(arg(i), resident=RES, next_images=arg(i + 1))
T["step"] = T["gather"] = 0.0
N = 500
t0 = time.perf_counter()
for i in range(50, 50 + N):
    fe.step(arg(i), resident=RES, next_images=arg(i + 1))
tot = time.perf_counter() - t0
print("us per step (resident %d): total %.1f | native step %.1f | gather+cross %.1f" % (RES, tot / N * 1e6, T["step"] / N * 1e6, T["gather"] / N * 1e6))
dist.destroy_process_group()
