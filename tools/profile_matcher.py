#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the matcher kernels (run under `rocprofv3 ... -- python3 tools/profile_matcher.py`).

Launches, on one stream: a 1 GiB hipMemset (known write bytes: WRITE_SIZE calibration), a 1 GiB device-to-device
copy (known read + write bytes: FETCH_SIZE calibration), then -- in their matrix-core and their popcount forms -- k_hamming_matrix at Q = R = 32000 and k_hamming_top2 at
4000 x 4000 and 32000 x 32000, a few launches each; then the radius-gated projection search of BASELINE.json configs[2]
(2 cameras 1280x720, 2000 features each, 4000 projected points: k_project + k_resolve) and the BoW row (k_bow_transform on
4000 descriptors of the stock-shape vocabulary, k_bow_join between two resident keyframes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import rt, synth

N = int(os.environ.get("MATRIX_N", "32000"))
GIB = 1 << 30
mt = m.Matcher()
st = mt.stream
a = rt.DeviceBuffer(GIB); b = rt.DeviceBuffer(GIB)
for _ in range(3):
    rt._L().orb_memset(a.ptr, 1, GIB, st)
    rt._L().orb_memcpy_d2d(b.ptr, a.ptr, GIB, st)
d = synth.descriptors(N, 4242)
dq = rt.DeviceBuffer(N * 32); dr = rt.DeviceBuffer(N * 32); dout = rt.DeviceBuffer(N * N * 2)
dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
res = [rt.DeviceBuffer(N * 4) for _ in range(3)]
sizes = []
for on in (1, 0):
    m.Matcher.use_matrix_cores(on)
    sizes += [m.Matcher.top2_scratch_bytes(N, N), m.Matcher.top2_scratch_bytes(4000, 4000)]
scr = rt.DeviceBuffer(max(sizes + [16]))
for on in (1, 0):  # matrix-core form (the default), then the xor/popcount form of the same two all-pairs kernels
    m.Matcher.use_matrix_cores(on)
    for _ in range(5):
        m.Matcher.hamming_matrix_device(dq.ptr, N, dr.ptr, N, dout.ptr, st)
    for _ in range(5):
        m.Matcher.hamming_top2_device(dq.ptr, 4000, dr.ptr, 4000, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
    for _ in range(5):
        m.Matcher.hamming_top2_device(dq.ptr, N, dr.ptr, N, res[0].ptr, res[1].ptr, res[2].ptr, scr.ptr, st)
m.Matcher.use_matrix_cores(-1)
rt.stream_sync(st)

# ---- cross-camera top-2 of configs[4] (8 cameras x 4000 descriptors in one list, own camera excluded), both forms
import multi_orb_slam_amd.matcher as mm
cams = [synth.perturbed_queries(d[:4000], 100 + c, 0.06) for c in range(8)]
frc = dict(un_x=np.zeros(32000, np.float32), un_y=np.zeros(32000, np.float32), octave=np.zeros(32000, np.int32), angle=np.zeros(32000, np.float32),
           uright=np.full(32000, -1, np.float32), cam_of=np.repeat(np.arange(8, dtype=np.int32), 4000),
           local_of=np.tile(np.arange(4000, dtype=np.int32), 8), descs=cams, bounds=(0.0, 0.0, 1920.0, 1080.0))
Fc = mt.frame(m.FrameData(**frc))
for on in (1, 0):
    m.Matcher.use_matrix_cores(on)
    for _ in range(5):
        bi, bd, sd = mt.cross_top2(Fc)
m.Matcher.use_matrix_cores(-1)
print("cross top-2 8 x 4000: %d features matched" % int((bi >= 0).sum()))
Fc.close()

# ---- configs[2]: SearchByProjection on a 2 x 2000-feature 1280x720 frame
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers  # noqa: E402  (input builders only; the oracle is not loaded here)
fr = helpers.make_frame_arrays([2000, 2000], 1280, 720, 7)
qs = helpers.make_queries(fr, 4000, 47, th=15.0)
F = mt.frame(m.FrameData(**fr))
for _ in range(5):
    n_match, _mo = mt.SearchByProjection(F, qs)
print("configs[2] projection search: %d matches" % n_match)
F.close()

# ---- BoW row
voc = synth.vocabulary(10, 6, seed=3)
V = m.Vocabulary(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], voc["L"])
S = m.BowSearch()
sides = []
for seed in (5, 6):
    f = synth.vocabulary_words(voc, 4000, seed=seed)
    for _ in range(5):
        (bow, fv) = V.bow_vectors(f, 4)
    ang = (helpers.rand_unit(4000, seed) * 360).astype(np.float32)
    sides.append(S.keyframe(m.BowSide(f, ang, fv)))
for _ in range(5):
    nm, _ = S.search_by_bow_resident(sides[0], sides[1], 1)
print("BoW search: %d matches" % nm)
print("done")
