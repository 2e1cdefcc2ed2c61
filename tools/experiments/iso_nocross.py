import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, rt
from multi_orb_slam_amd.frontend import SKIP_CROSS
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
dev = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(8)]
for t in range(8):
    for c in range(2):
        dev[t][c].upload(synth.image(c, t, W, H))
ts = []
for it in range(60):
    t0 = time.perf_counter()
    imgs = [(dev[it % 8][c].ptr, W, H, W, 1) for c in range(2)]
    r = fe.fe.step(imgs, None, SKIP_CROSS, copy=False, motion=(pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ))
    ts.append(time.perf_counter() - t0)
    time.sleep(0.002)
ts = sorted(ts[10:])
print(json.dumps({"isolated_nocross_us_median": round(1e6 * ts[len(ts) // 2], 1)}))
fe.close()
