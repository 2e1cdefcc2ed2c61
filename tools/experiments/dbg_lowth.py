import os, sys, faulthandler
faulthandler.enable()
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
import oracle
w, h = 640, 480
which = sys.argv[1] if len(sys.argv) > 1 else "both"
kw = dict(nfeatures=1000, ini_th_fast=12, min_th_fast=3)
p = m.ExtractorParams(**kw)
imgs = [synth.image(7 + c, 0, w, h) for c in range(2)]
if which in ("oracle_first", "both"):
    ok, od = oracle.extract(imgs[0], nfeatures=1000, ini_th=12, min_th=3); print("oracle first", len(ok), flush=True)
if which != "oracle_only":
    ex = m.Extractor([p, p], w, h)
    print("created", flush=True)
    out = ex.extract(imgs)
    print("product", [len(o[0]) for o in out], "path", ex.last_path(), flush=True)
ok, od = oracle.extract(imgs[0], nfeatures=1000, ini_th=12, min_th=3); print("oracle after", len(ok), flush=True)
if which != "oracle_only":
    k, d = out[0]
    print("equal", k.tobytes() == ok.tobytes() and np.array_equal(d, od), flush=True)
