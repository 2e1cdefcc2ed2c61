"""The drop-in class path, timed and checked: host/test_host `dropin` runs the reference's per-frame call pattern through the
C++ classes that keep the reference's signatures -- two ORBextractor::operator() calls back to back (src/Frame.cc:182,185;
or one ExtractBatch), the reference's host-side Frame assembly, then a stack-constructed ORBmatcher(0.9, true) and
SearchByProjection(CurrentFrame, LastFrame, 15, false, Calib) (src/Tracking.cc:1237-1267) -- on the synthetic stream.
TEST / BENCH INFRASTRUCTURE: the checker half compares the last step with the CPU oracle."""
import os
import struct
import subprocess
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "multi_orb_slam_amd", "host", "test_host")
f32 = np.float32
INTR = (f32(520.0), f32(520.0), f32(320.0), f32(240.0))   # fx, fy, cx, cy of the synthetic rig


def run(width=640, height=480, nf=(1000, 500), T=12, iters=200, warmup=30, batch=False, check=True, workdir=None, frames=None):
    """-> dict(us medians per call, class_calls_us, fps, per-step p5/p95, parity).
    T (frames of the synthetic ring) is larger than the per-thread cache of uploaded frames (8, least recently used), so
    every timed SearchByProjection uploads its current frame as a live stream would: the timing never rides on a cache hit.
    frames: [T][2] uint8 images moving by pipeline.MOTION per step in place of the synthetic stream (tests/natural.py)."""
    from multi_orb_slam_amd import synth, pipeline
    import oracle
    tmp = workdir or tempfile.mkdtemp(prefix="morb_dropin_")
    if frames is None:
        frames = [[synth.image(c, t, width, height) for c in range(2)] for t in range(T)]
    assert len(frames) == T
    depth = [pipeline.synth_depth_image(c, width, height) for c in range(2)]
    fx, fy, cx, cy = INTR
    blob = struct.pack("<7i", width, height, nf[0], nf[1], T, iters, warmup)
    blob += struct.pack("<8f", fx, fy, cx, cy, pipeline.MBF, pipeline.TH_PROJ, pipeline.MOTION[0], pipeline.MOTION[1])
    blob += b"".join(frames[t][c].tobytes() for t in range(T) for c in range(2))
    blob += b"".join(d.astype(np.float32).tobytes() for d in depth)
    sp, op = os.path.join(tmp, "stream.bin"), os.path.join(tmp, "out.bin")
    with open(sp, "wb") as fh:
        fh.write(blob)
    subprocess.check_call([BIN, "dropin", sp, op, "1" if batch else "0"], timeout=180)
    buf = open(op, "rb").read()
    meds = np.frombuffer(buf, np.float32, 14, 0)
    per = np.frombuffer(buf, np.float32, iters, 56)[1:]        # (step 0 has no last frame to search)
    off = 56 + 4 * iters
    assert meds[11] == 0, "%d class calls failed" % int(meds[11])
    out = {"pattern": "ExtractBatch + SearchByProjection" if batch else "2 x operator() + SearchByProjection",
           "extract_cam0_us": round(float(meds[0]), 1), "extract_cam1_us": round(float(meds[1]), 1),
           "reference_host_frame_assembly_us": round(float(meds[2]), 1), "search_by_projection_us": round(float(meds[3]), 1),
           "search_breakdown_us": {"reference_host_projection": round(float(meds[6]), 1), "frame_hash_and_upload": round(float(meds[7]), 1),
                                   "device_search": round(float(meds[8]), 1)},
           "frame_cache": {"hits": int(meds[9]), "misses": int(meds[10])},
           "descriptors_found_in_hbm": {"served": int(meds[12]), "sent_from_host": int(meds[13])},
           "class_calls_us": round(float(np.median(per)), 1), "class_calls_p5_us": round(float(np.percentile(per, 5)), 1),
           "class_calls_p95_us": round(float(np.percentile(per, 95)), 1), "loop_us": round(float(meds[5]), 1),
           "dropin_fps": round(1e6 / float(np.median(per)), 1), "steps": int(iters)}
    if not check:
        return out
    # ---- checker: the last step against the oracle
    t_last = (warmup + iters - 1) % T
    t_prev = (warmup + iters - 2) % T
    got = []
    for c in range(2):
        n = struct.unpack_from("<i", buf, off)[0]; off += 4
        k = np.frombuffer(buf, oracle.KP_DTYPE, n, off).copy(); off += 28 * n
        d = np.frombuffer(buf, np.uint8, n * 32, off).reshape(n, 32).copy(); off += 32 * n
        ok, od = oracle.extract(frames[t_last][c], nfeatures=nf[c])
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od), "drop-in extraction differs from the oracle (camera %d)" % c
        got.append((k, d))
    nmatches = struct.unpack_from("<i", buf, off)[0]; off += 4
    n_cur = len(got[0][0]) + len(got[1][0])
    match_src = np.frombuffer(buf, np.int32, n_cur, off); off += 4 * n_cur
    nl, nl0 = struct.unpack_from("<ii", buf, off); off += 8
    world = np.frombuffer(buf, np.float32, 3 * nl, off).reshape(nl, 3); off += 12 * nl
    loct = np.frombuffer(buf, np.int32, nl, off); off += 4 * nl
    lang = np.frombuffer(buf, np.float32, nl, off); off += 4 * nl
    ldesc = np.frombuffer(buf, np.uint8, nl * 32, off).reshape(nl, 32); off += 32 * nl
    # the last frame the driver matched from must itself be the oracle's extraction of the previous images
    prev = [oracle.extract(frames[t_prev][c], nfeatures=nf[c]) for c in range(2)]
    assert nl0 == len(prev[0][0]) and nl == nl0 + len(prev[1][0])
    assert np.array_equal(ldesc, np.concatenate([p[1] for p in prev])) and np.array_equal(loct, np.concatenate([p[0]["octave"] for p in prev]))
    # queries: the host-side projection of src/ORBmatcher.cc:3502-3552 in float32 (identity poses)
    from multi_orb_slam_amd._lib import QUERY_DTYPE
    scale = oracle.tables()["scale"]
    mbf, th = f32(pipeline.MBF), f32(pipeline.TH_PROJ)
    q = np.zeros(nl, QUERY_DTYPE); src = []
    for i in range(nl):
        w3 = world[i]
        invz = f32(1.0 / np.float64(w3[2]))
        if invz < 0:
            continue
        u = f32(f32(f32(fx * w3[0]) * invz) + cx); v = f32(f32(f32(fy * w3[1]) * invz) + cy)
        if u < 0 or u > width or v < 0 or v > height:
            continue
        o = int(loct[i]); e = q[len(src)]
        e["u"] = u; e["v"] = v; e["radius"] = f32(th * scale[o]); e["ur"] = f32(u - f32(mbf * invz))
        e["min_level"] = o - 1; e["max_level"] = o + 1; e["cam"] = 0 if i < nl0 else 1; e["blocks"] = 1
        e["angle"] = lang[i]; e["desc"] = ldesc[i]
        src.append(i)
    q = q[:len(src)].copy(); src = np.array(src)
    kps = np.concatenate([g[0] for g in got]); counts = [len(g[0]) for g in got]
    offs = np.concatenate([[0], np.cumsum(counts)])
    st = [oracle.stereo_from_depth(g[0], depth[c], pipeline.MBF, kps["x"][offs[c]:offs[c + 1]]) for c, g in enumerate(got)]
    fr = oracle.FrameData(kps["x"], kps["y"], kps["octave"], kps["angle"], np.concatenate([a for a, _ in st]),
                          np.repeat(np.arange(2, dtype=np.int32), counts), np.concatenate([np.arange(n, dtype=np.int32) for n in counts]),
                          [g[1] for g in got], (0.0, 0.0, float(width), float(height)))
    en, emo = oracle.search_by_projection_frames(fr, q, 100, True)
    exp = np.where(emo >= 0, src[np.maximum(emo, 0)], -1)
    assert nmatches == en and np.array_equal(match_src, exp), "drop-in SearchByProjection differs from the oracle"
    assert en > 100
    out["parity"] = "last step bit-exact vs oracle: both cameras' keypoints + descriptors, %d SearchByProjection matches" % en
    return out


def bench(width, height, nfeat):
    """bench.py leg: both integration patterns, cameras as the reference configures them (camera 2 gets nFeatures/2,
    src/Tracking.cc:144-145) and with nfeat on both (the configs[1] workload)."""
    res = {}
    for name, nf, batch in (("reference_pattern", (nfeat, nfeat), False), ("batched_extract", (nfeat, nfeat), True)):
        res[name] = run(width, height, nf, batch=batch)
        assert res[name]["frame_cache"]["hits"] == 0      # every timed search uploaded its frame
    res["dropin_fps"] = res["reference_pattern"]["dropin_fps"]
    return res
