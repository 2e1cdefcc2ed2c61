"""The C++ host classes with the reference's signatures (multi_orb_slam_amd/host: ORB_SLAM2::ORBextractor,
ORB_SLAM2::ORBmatcher) driven through host/test_host on the GPU, compared bit-for-bit with the oracle."""
import os
import struct
import subprocess
import numpy as np
import pytest
import helpers
from multi_orb_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multi_orb_slam_amd", "host")
BIN = os.path.join(HOST, "test_host")
f32 = np.float32


def test_host_library_exports_reference_signatures():
    """CPU-side: the C++ wrapper library exists and exports the reference's public methods."""
    so = os.path.join(ROOT, "multi_orb_slam_amd", "lib", "libmorb_host.so")
    assert os.path.exists(so) and os.path.exists(BIN), "run __graft_entry__.build()"
    syms = subprocess.check_output(["nm", "-DC", so]).decode()
    for want in ("ORB_SLAM2::ORBextractor::ORBextractor(int, float, int, int, int)",
                 "ORB_SLAM2::ORBextractor::operator()(cv::Mat const&, cv::Mat const&, std::vector<cv::KeyPoint",
                 "ORB_SLAM2::ORBmatcher::ORBmatcher(float, bool)",
                 "ORB_SLAM2::ORBmatcher::DescriptorDistance(cv::Mat const&, cv::Mat const&)",
                 "ORB_SLAM2::ORBmatcher::SearchByProjection(ORB_SLAM2::Frame&, std::vector<ORB_SLAM2::MapPoint*",
                 "ORB_SLAM2::ORBmatcher::SearchByProjection(ORB_SLAM2::Frame&, ORB_SLAM2::Frame const&, float, bool, cv::Mat)",
                 "ORB_SLAM2::ORBmatcher::TH_HIGH", "ORB_SLAM2::ORBmatcher::TH_LOW", "ORB_SLAM2::ORBmatcher::HISTO_LENGTH"):
        assert want in syms, want


def _read_kps(buf, off):
    import oracle
    n = struct.unpack_from("<i", buf, off)[0]; off += 4
    k = np.frombuffer(buf, oracle.KP_DTYPE, n, off).copy(); off += 28 * n
    d = np.frombuffer(buf, np.uint8, n * 32, off).reshape(n, 32).copy(); off += 32 * n
    return k, d, off


@pytest.mark.gpu
def test_cpp_orbextractor_operator_call(tmp_path):
    import oracle
    img = synth.image(0, 3, 640, 480)
    (tmp_path / "img.bin").write_bytes(img.tobytes())
    subprocess.check_call([BIN, "extract", str(tmp_path / "img.bin"), "640", "480", "1000", str(tmp_path / "out.bin")])
    buf = (tmp_path / "out.bin").read_bytes()
    k, d, off = _read_kps(buf, 0)
    ok, od = oracle.extract(img, nfeatures=1000)
    assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)
    sf = np.frombuffer(buf, np.float32, 8, off)
    assert np.array_equal(sf, oracle.tables()["scale"])


@pytest.mark.gpu
def test_cpp_orbextractor_batch_two_cameras(tmp_path):
    import oracle
    imgs = [synth.image(c, 1, 640, 480) for c in range(2)]
    for c in range(2):
        (tmp_path / ("img%d.bin" % c)).write_bytes(imgs[c].tobytes())
    subprocess.check_call([BIN, "batch", str(tmp_path / "out.bin"), "640", "480", str(tmp_path / "img0.bin"), "1000",
                           str(tmp_path / "img1.bin"), "500"])   # cam 2 gets nFeatures/2 (reference Tracking.cc:145)
    buf = (tmp_path / "out.bin").read_bytes()
    off = 0
    for c, nf in enumerate((1000, 500)):
        k, d, off = _read_kps(buf, off)
        ok, od = oracle.extract(imgs[c], nfeatures=nf)
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)


def _frame_bytes(fr, n0, n1, scale, Tcw, intr, bounds):
    out = struct.pack("<ii", n0, n1)
    for key in ("un_x", "un_y", "angle", "uright"):
        out += np.asarray(fr[key], np.float32).tobytes()
    out += np.asarray(fr["octave"], np.int32).tobytes()
    out += fr["descs"][0].tobytes() + fr["descs"][1].tobytes()
    out += np.asarray(scale, np.float32).tobytes() + np.asarray(Tcw, np.float32).tobytes()
    out += np.asarray(intr, np.float32).tobytes() + np.asarray(bounds, np.float32).tobytes()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("check_ori", [1, 0])
def test_cpp_orbmatcher_search_by_projection_overloads(tmp_path, check_ori):
    import oracle
    from multi_orb_slam_amd._lib import QUERY_DTYPE
    W, H = 640, 480
    fx, fy, cx, cy, mbf = f32(520.9), f32(521.0), f32(325.1), f32(249.7), f32(40.0)
    scale = oracle.tables()["scale"]
    n0, n1 = 700, 400
    cur = helpers.make_frame_arrays([n0, n1], W, H, 31)
    last = helpers.make_frame_arrays([n0 - 50, n1 + 30], W, H, 32)
    nl = len(last["un_x"]); nl0 = n0 - 50
    # map points of the last frame: placed so that they project (identity pose) near current-frame features
    pick = (helpers.rand_u32(nl, 77) % len(cur["un_x"])).astype(np.int64)
    # keep the camera: last-frame cam-0 points look at cam-0 features
    lcam = np.array(last["cam_of"]); ccam = np.array(cur["cam_of"])
    for i in range(nl):
        while ccam[pick[i]] != lcam[i]:
            pick[i] = (pick[i] + 1) % len(ccam)
    z = (f32(1.5) + (helpers.rand_unit(nl, 78) * 6).astype(np.float32)).astype(np.float32)
    tu = (cur["un_x"][pick] + ((helpers.rand_unit(nl, 79) - 0.5) * 8).astype(np.float32)).astype(np.float32)
    tv = (cur["un_y"][pick] + ((helpers.rand_unit(nl, 80) - 0.5) * 8).astype(np.float32)).astype(np.float32)
    xc = ((tu - cx) / fx * z).astype(np.float32); yc = ((tv - cy) / fy * z).astype(np.float32)
    t12 = np.array([0.1, 0.0, 0.0], np.float32)
    world = np.stack([xc, yc, z], 1).astype(np.float32)
    world[lcam == 1] = (world[lcam == 1] + t12).astype(np.float32)   # cam 2: x3Dc = I * x3Dw + (-t)
    alld = np.concatenate(cur["descs"])
    mp_desc = synth.perturbed_queries(alld[pick], 5, 0.05)
    mp_desc[::2] = alld[pick][::2]
    obs = np.ones(nl, np.int32); has = (helpers.rand_unit(nl, 81) < 0.9).astype(np.int32)
    outl = (helpers.rand_unit(nl, 82) < 0.05).astype(np.int32)
    last["octave"] = cur["octave"][pick].astype(np.int32)
    last["angle"] = np.mod(cur["angle"][pick] + f32(2.0), f32(360.0)).astype(np.float32)   # consistent rotation
    th = f32(15.0)
    eye = np.eye(4, dtype=np.float32)
    blob = _frame_bytes(cur, n0, n1, scale, eye, (fx, fy, cx, cy, mbf), (0, 0, W, H))
    blob += _frame_bytes(last, nl0, nl - nl0, scale, eye, (fx, fy, cx, cy, mbf), (0, 0, W, H))
    for i in range(nl):
        blob += world[i].tobytes() + mp_desc[i].tobytes() + struct.pack("<iii", int(obs[i]), int(has[i]), int(outl[i]))
    calib = np.concatenate([np.eye(3, dtype=np.float32).ravel(), t12])
    blob += calib.tobytes() + struct.pack("<fi", float(th), check_ori)
    # local map points for the second overload
    nloc = 600
    lp = (helpers.rand_u32(nloc, 90) % n0).astype(np.int64)
    lvl = np.maximum(cur["octave"][lp], 0).astype(np.int32)
    px = (cur["un_x"][lp] + ((helpers.rand_unit(nloc, 91) - 0.5) * 4).astype(np.float32)).astype(np.float32)
    py = (cur["un_y"][lp] + ((helpers.rand_unit(nloc, 92) - 0.5) * 4).astype(np.float32)).astype(np.float32)
    pxr = (px - f32(20.0)).astype(np.float32)
    vcos = np.where(helpers.rand_unit(nloc, 93) < 0.5, f32(0.9995), f32(0.9)).astype(np.float32)
    ldesc = synth.perturbed_queries(alld[lp], 6, 0.05); ldesc[::3] = alld[lp][::3]
    inview = (helpers.rand_unit(nloc, 94) < 0.9).astype(np.int32); bad = (helpers.rand_unit(nloc, 95) < 0.05).astype(np.int32)
    blob += struct.pack("<i", nloc)
    for i in range(nloc):
        blob += struct.pack("<fffif", float(px[i]), float(py[i]), float(pxr[i]), int(lvl[i]), float(vcos[i]))
        blob += ldesc[i].tobytes() + struct.pack("<ii", int(inview[i]), int(bad[i]))
    th2 = f32(3.0)
    blob += struct.pack("<f", float(th2))
    (tmp_path / "case.bin").write_bytes(blob)
    subprocess.check_call([BIN, "match", str(tmp_path / "case.bin"), str(tmp_path / "out.bin")])
    buf = (tmp_path / "out.bin").read_bytes()
    n_total = n0 + n1
    got_n1 = struct.unpack_from("<i", buf, 0)[0]
    got_m1 = np.frombuffer(buf, np.int32, n_total, 4)
    got_n2 = struct.unpack_from("<i", buf, 4 + 4 * n_total)[0]
    got_m2 = np.frombuffer(buf, np.int32, n0, 8 + 4 * n_total)
    got_dd = struct.unpack_from("<i", buf, 8 + 4 * n_total + 4 * n0)[0]

    # ---- expected, overload 1: the host-side projection of src/ORBmatcher.cc:3502-3552 in float32, then the oracle
    q = []; src = []
    for i in range(nl):
        if not has[i] or outl[i]:
            continue
        w = world[i]
        if lcam[i] == 1:
            w = (w + (-t12)).astype(np.float32)
        invz = f32(1.0 / np.float64(w[2]))
        if invz < 0:
            continue
        u = f32(f32(f32(fx * w[0]) * invz) + cx); v = f32(f32(f32(fy * w[1]) * invz) + cy)
        if u < 0 or u > W or v < 0 or v > H:
            continue
        o = int(last["octave"][i])
        e = np.zeros(1, QUERY_DTYPE)
        e["u"] = u; e["v"] = v; e["radius"] = f32(th * scale[o]); e["ur"] = f32(u - f32(mbf * invz))
        e["min_level"] = o - 1; e["max_level"] = o + 1; e["cam"] = int(lcam[i]); e["blocks"] = 1
        e["angle"] = last["angle"][i]; e["desc"] = mp_desc[i]
        q.append(e); src.append(i)
    q = np.concatenate(q); src = np.array(src)
    OF = oracle.FrameData(**cur)
    en1, emo = oracle.search_by_projection_frames(OF, q, 100, bool(check_ori))
    exp1 = np.where(emo >= 0, src[np.maximum(emo, 0)], -1)
    assert got_n1 == en1 and np.array_equal(got_m1, exp1)
    assert got_n1 > 100

    # ---- expected, overload 2 (src/ORBmatcher.cc:62-149)
    q2 = []; src2 = []
    for i in range(nloc):
        if not inview[i] or bad[i]:
            continue
        r = f32(2.5) if vcos[i] > 0.998 else f32(4.0)
        r = f32(r * th2)
        e = np.zeros(1, QUERY_DTYPE)
        e["u"] = px[i]; e["v"] = py[i]; e["radius"] = f32(r * scale[lvl[i]]); e["ur"] = pxr[i]
        e["min_level"] = lvl[i] - 1; e["max_level"] = lvl[i]; e["cam"] = 0; e["blocks"] = 1; e["desc"] = ldesc[i]
        q2.append(e); src2.append(i)
    q2 = np.concatenate(q2); src2 = np.array(src2)
    en2, emo2 = oracle.search_by_projection_points(OF, q2, None, 0.8, 100)
    exp2 = np.where(emo2[:n0] >= 0, src2[np.maximum(emo2[:n0], 0)], -1)
    assert got_n2 == en2 and np.array_equal(got_m2, exp2)
    assert got_n2 > 50
    assert got_dd == oracle.descriptor_distance(cur["descs"][0][0], cur["descs"][0][1])
