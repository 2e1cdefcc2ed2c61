import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt, synth
from oracle_pipeline import OracleFrontEnd, assert_same_step
shapes = [(752, 480, (1200,)), (400, 300, (300, 300)), (640, 480, (1000, 1000, 1000, 1000)), (500, 375, (777, 333)), (1024, 768, (1500, 1500)),
          (800, 250, (600, 600)), (256, 256, (200, 200)), (640, 480, (2000, 50))]
for (w, h, nfs) in shapes:
    params = [m.ExtractorParams(nfeatures=n) for n in nfs]
    fe = pipeline.FrontEnd(params, w, h); ofe = OracleFrontEnd(params, w, h)
    keep = []
    for t in range(4):
        imgs = [synth.image(c + 3, t, w, h) for c in range(len(nfs))]
        if t % 2:
            row = []
            for im in imgs:
                b = rt.DeviceBuffer(im.nbytes); b.upload(im); row.append(b)
            keep.append(row)
            got = fe.step([(b.ptr, w) for b in row], resident=True)
        else:
            got = fe.step(imgs)
        assert_same_step(got, ofe.step(imgs))
    print("ok", w, h, nfs, got["counts"], got["n_temporal"], fe.ex.debug_last_path() if hasattr(fe.ex, "debug_last_path") else "")
    fe.close()
print("ALL OK")
