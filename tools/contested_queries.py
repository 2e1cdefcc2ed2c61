#!/usr/bin/env python3
"""How many SearchByProjection queries of the benchmark stream share a candidate feature with another query?  (A query
whose candidates nobody else lists is decided by its first evaluation; only the others need the resolve's sweeps.)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, pipeline, _lib
W, H = 640, 480
fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
for t in range(6):
    r = fe.step([synth.image(c, t, W, H) for c in range(2)])
res = fe.fe._res
nq = res.n_queries
q = np.frombuffer((C.c_char * (nq * 68)).from_address(C.cast(res.queries, C.c_void_p).value), dtype=_lib.QUERY_DTYPE).copy()
n = r["kps"].shape[0]
x, y, octv = np.asarray(r["un_x"]), np.asarray(r["un_y"]), np.asarray(r["kps"]["octave"])
cam = np.repeat(np.arange(2), r["counts"])
lists = []
for i in range(nq):
    lo, hi = q["min_level"][i], q["max_level"][i]
    ok = (cam == q["cam"][i]) & (np.abs(x - q["u"][i]) < q["radius"][i]) & (np.abs(y - q["v"][i]) < q["radius"][i]) & (octv >= lo)
    if hi >= 0:
        ok &= octv <= hi
    lists.append(np.nonzero(ok)[0])
cnt = np.zeros(n, np.int64)
for l in lists:
    cnt[l] += 1
free = sum(1 for l in lists if len(l) == 0 or (cnt[l] == 1).all())
lens = np.array([len(l) for l in lists])
print("queries %d, features %d; candidates per query: median %d, p95 %d, max %d; queries sharing no candidate: %d (%.0f %%); empty lists %d"
      % (nq, n, np.median(lens), np.percentile(lens, 95), lens.max(), free, 100.0 * free / nq, (lens == 0).sum()))
# shortlists only (the K best by distance are what the sweeps walk): an upper bound of the sharing uses the full lists above
