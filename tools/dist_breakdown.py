#!/usr/bin/env python3
"""bench.py's forced-exchange loop with per-piece host timers (one GPU, world-1 RCCL group)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import torch, torch.distributed as dist
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.dist import DescriptorExchange
dist.init_process_group("nccl", rank=0, world_size=1)
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
arg = lambda t: [(dev[t % RING][c].ptr, W) for c in range(2)]
fe.copy_results = False
T = {}


def timed(obj, name, key):
    f = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)


MODE = os.environ.get("DB_MODE", "")
if MODE == "nocross":      # the collective only: no gathered matching (results are then stale; timing experiment)
    fe.gather.enqueue = lambda f: (setattr(fe.gather, "_ahead", None))
    fe.gather.collect = lambda f, views=False: (None, None, None, [0] * 2)
    import multi_orb_slam_amd.pipeline as _pl
    def _step(images, resident=False, next_images=None):
        if next_images is not None: fe.announce(next_images, resident)
        images = [(im[0], fe.width, fe.height, im[1], 1) for im in images]
        ahead = fe.fe.peek_block(images)
        if ahead is not None: fe.gather.gather_ahead(fe, ahead)
        fe.fe.begin(images, None, 1, motion=(_pl.MOTION[0], _pl.MOTION[1], _pl.TH_PROJ))
        fe.gather._ahead = None
        return fe.fe.end(copy=False)
    fe.step = _step
if MODE == "noorder":      # gathered matching not ordered behind the collective's stream (timing experiment only)
    _orig = fe.mt.cross_top2_gathered_enqueue
    fe.mt.cross_top2_gathered_enqueue = lambda *a: _orig(*a[:6], None)
timed(fe.fe, "begin", "begin"); timed(fe.fe, "end", "end"); timed(fe.fe, "prefetch", "prefetch")
timed(fe.gather, "enqueue", "exchange.enqueue"); timed(fe.gather, "collect", "exchange.collect"); timed(fe.gather, "_gather", "  of which all_gather call")
g_call = fe.gather.__call__
class G:
    def __getattr__(self, k): return getattr(fe_gather, k)
fe_gather = fe.gather
gc.collect(); gc.freeze(); gc.disable()
fe.announce(arg(1), resident=True)
for i in range(100):
    fe.step(arg(i), resident=True, next_images=arg(i + 2))
T.clear(); fe.early_exchanges = 0
N = 1000
t0 = time.perf_counter()
for i in range(100, 100 + N):
    fe.step(arg(i), resident=True, next_images=arg(i + 2))
tot = time.perf_counter() - t0
print("us per step: total %.1f; early exchanges %d of %d" % (tot / N * 1e6, fe.early_exchanges, N))
for k, v in T.items():
    print("  %-28s %.1f" % (k, v / N * 1e6))
dist.destroy_process_group()
