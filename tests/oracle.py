"""ctypes binding of oracle/liborb_oracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
# (MORB_ORACLE_SO: another build of the same sources, e.g. oracle/_native/liborb_oracle.so = `make -C oracle native`, -march=native,
# which bench.py's cpu_baseline leg times in a child process next to the portable build)
_SO = os.environ.get("MORB_ORACLE_SO") or os.path.join(ORACLE_DIR, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class Query(C.Structure):
    _fields_ = [("u", C.c_float), ("v", C.c_float), ("radius", C.c_float), ("ur", C.c_float),
                ("min_level", C.c_int), ("max_level", C.c_int), ("cam", C.c_int), ("blocks", C.c_int),
                ("angle", C.c_float), ("desc", C.c_uint8 * 32)]


QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("min_level", "<i4"),
                        ("max_level", "<i4"), ("cam", "<i4"), ("blocks", "<i4"), ("angle", "<f4"),
                        ("desc", "u1", (32,))])
assert QUERY_DTYPE.itemsize == C.sizeof(Query) == 68


class Frame(C.Structure):
    _fields_ = [("n_total", C.c_int), ("n_cams", C.c_int), ("un_x", C.c_void_p), ("un_y", C.c_void_p),
                ("octave", C.c_void_p), ("angle", C.c_void_p), ("uright", C.c_void_p), ("cam_of", C.c_void_p),
                ("local_of", C.c_void_p), ("desc", C.c_void_p), ("minX", C.c_float), ("minY", C.c_float),
                ("maxX", C.c_float), ("maxY", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liborb_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            if os.environ.get("MORB_ORACLE_SO"):
                raise FileNotFoundError(_SO)
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_ic_angle.restype = C.c_float
        _lib.orc_det_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _lib.orc_orb_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_tables.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        _lib.orc_level_sizes.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        _lib.orc_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                              C.c_void_p, C.c_int]
        _lib.orc_search_by_projection_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                         C.c_void_p]
        _lib.orc_search_by_projection_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                                         C.c_int, C.c_void_p]
        _lib.orc_search_by_projection_loop2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_project_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_vocab_create.restype = C.c_void_p
        _lib.orc_vocab_create.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4
        _lib.orc_vocab_destroy.argtypes = [C.c_void_p]
        _lib.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_bow_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
        _lib.orc_bow_score_l1.restype = C.c_double
        _lib.orc_bow_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        _lib.orc_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


# ---------------------------------------------------------------------------------------------- tables
def tables(nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
    L = lib()
    sc = np.zeros(nlevels, np.float32); inv = np.zeros_like(sc); s2 = np.zeros_like(sc); is2 = np.zeros_like(sc)
    quota = np.zeros(nlevels, np.int32); umax = np.zeros(16, np.int32)
    L.orc_tables(nfeatures, scale_factor, nlevels, ini_th, min_th, _p(sc), _p(inv), _p(s2), _p(is2), _p(quota), _p(umax))
    return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)


def level_sizes(W, H, scale_factor=1.2, nlevels=8):
    w = np.zeros(nlevels, np.int32); h = np.zeros(nlevels, np.int32)
    lib().orc_level_sizes(W, H, scale_factor, nlevels, _p(w), _p(h))
    return [(int(a), int(b)) for a, b in zip(w, h)]


def pattern():
    out = np.zeros(1024, np.int8)
    lib().orc_pattern(_p(out))
    return out.reshape(256, 4)


# ---------------------------------------------------------------------------------------------- image ops
def resize_linear(src, dw, dh):
    src = _u8(src); sh, sw = src.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    return dst


def copy_make_border(src, border):
    src = _u8(src); h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_copy_make_border_reflect101(_p(src), w, h, w, _p(dst), border, w + 2 * border)
    return dst


def pyramid(img, scale_factor=1.2, nlevels=8):
    img = _u8(img); H, W = img.shape
    sizes = level_sizes(W, H, scale_factor, nlevels)
    out = np.zeros(sum(w * h for w, h in sizes), np.uint8)
    lib().orc_pyramid(_p(img), W, H, W, scale_factor, nlevels, _p(out))
    levels, off = [], 0
    for w, h in sizes:
        levels.append(out[off:off + w * h].reshape(h, w).copy()); off += w * h
    return levels


def corner_score(img, x, y, threshold):
    img = _u8(img)
    return lib().orc_corner_score(_p(img), img.shape[1], x, y, threshold)


def is_corner(img, x, y, threshold):
    img = _u8(img)
    return bool(lib().orc_is_corner(_p(img), img.shape[1], x, y, threshold))


def fast(view, threshold, cap=100000):
    view = _u8(view); rows, cols = view.shape
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_fast(_p(view), cols, rows, cols, threshold, _p(out), cap)
    assert n <= cap
    return out[:n]


def cell_candidates(img, ini_th=20, min_th=7, cap=400000):
    img = _u8(img); h, w = img.shape
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_cell_candidates(_p(img), w, h, ini_th, min_th, _p(out), cap)
    assert n <= cap
    return out[:n]


def distribute_octree(kps, minX, maxX, minY, maxY, N):
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    out = np.zeros(max(len(kps), 1), KP_DTYPE)
    n = lib().orc_distribute_octree(_p(kps), len(kps), minX, maxX, minY, maxY, N, _p(out), len(out))
    if n == -2 ** 31:
        raise ValueError("the reference's DistributeOctTree is undefined for a %d x %d region (round(width / height) == 0)" % (maxX - minX, maxY - minY))
    return out[:n]


def fast_atan2(y, x):
    return np.float32(lib().orc_fast_atan2(float(y), float(x)))


def ic_angle(img, x, y):
    img = _u8(img); h, w = img.shape
    m01 = C.c_int(); m10 = C.c_int()
    a = lib().orc_ic_angle(_p(img), w, h, int(x), int(y), C.byref(m01), C.byref(m10))
    return np.float32(a), m01.value, m10.value


def gaussian_kernel():
    k = np.zeros(7, np.int32); lib().orc_gaussian_kernel(_p(k)); return k


def gaussian_blur7(img):
    img = _u8(img); h, w = img.shape
    dst = np.zeros_like(img)
    lib().orc_gaussian_blur7(_p(img), w, h, _p(dst))
    return dst


def det_sincos(angle_rad):
    c = C.c_float(); s = C.c_float()
    lib().orc_det_sincos(float(angle_rad), C.byref(c), C.byref(s))
    return np.float32(c.value), np.float32(s.value)


def orb_descriptor(blurred, x, y, angle_deg):
    blurred = _u8(blurred); h, w = blurred.shape
    d = np.zeros(32, np.uint8)
    lib().orc_orb_descriptor(_p(blurred), w, h, float(x), float(y), float(angle_deg), _p(d))
    return d


def extract(img, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
    img = _u8(img); H, W = img.shape
    cap = nfeatures + 3 * nlevels + 64
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    n = lib().orc_extract(_p(img), W, H, W, nfeatures, scale_factor, nlevels, ini_th, min_th, _p(kps), _p(desc), cap)
    if n == -2 ** 31:
        raise ValueError("the reference's DistributeOctTree is undefined for a %d x %d image (round(width / height) == 0 on some level)" % (W, H))
    assert n >= 0, "oracle capacity"
    return kps[:n].copy(), desc[:n].copy()


# ---------------------------------------------------------------------------------------------- matcher
def descriptor_distance(a, b):
    a = _u8(a); b = _u8(b)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def stereo_from_depth(kps, depth, mbf, un_x=None):
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    ur = np.zeros(len(kps), np.float32); dd = np.zeros(len(kps), np.float32)
    if un_x is not None:
        un_x = np.ascontiguousarray(un_x, np.float32)
    L = lib()
    L.orc_stereo_from_depth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_stereo_from_depth(_p(kps), len(kps), _p(depth), depth.shape[1], mbf, _p(ur), _p(dd), _p(un_x) if un_x is not None else None)
    return ur, dd


def calib_array(calib):
    """(fx, fy, cx, cy, k1, k2, p1, p2[, k3]) -> float32[9] (the reference holds K and DistCoef as CV_32F)."""
    c = list(calib) + [0.0] * (9 - len(calib))
    return np.array(c, np.float32)


def undistort_points(calib, x, y):
    """Frame::UndistortKeyPoints: cv::undistortPoints(pts, K, dist, noArray(), K); a copy when k1 == 0 or calib is None."""
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    ux = np.zeros_like(x); uy = np.zeros_like(y)
    c = calib_array(calib) if calib is not None else None
    L = lib()
    L.orc_undistort_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_undistort_points(_p(c) if c is not None else None, _p(x), _p(y), len(x), _p(ux), _p(uy))
    return ux, uy


def image_bounds(calib, cols, rows):
    """Frame::ComputeImageBounds -> (minX, minY, maxX, maxY)."""
    out = np.zeros(4, np.float32)
    c = calib_array(calib) if calib is not None else None
    L = lib()
    L.orc_image_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_image_bounds(_p(c) if c is not None else None, cols, rows, _p(out))
    return tuple(float(v) for v in out)


def bf_top2(q, r):
    q = _u8(q); r = _u8(r)
    bi = np.zeros(len(q), np.int32); bd = np.zeros(len(q), np.int32); sd = np.zeros(len(q), np.int32)
    lib().orc_bf_top2(_p(q), len(q), _p(r), len(r), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def hamming_matrix(q, r):
    q = _u8(q); r = _u8(r)
    out = np.zeros((len(q), len(r)), np.uint16)
    lib().orc_hamming_matrix(_p(q), len(q), _p(r), len(r), _p(out))
    return out


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32); ind = np.zeros(3, np.int32)
    lib().orc_three_maxima(_p(sizes), len(sizes), _p(ind))
    return tuple(int(i) for i in ind)


class FrameData:
    """Flat arrays of the Frame members the matcher reads, cam-major global indexing."""

    def __init__(self, un_x, un_y, octave, angle, uright, cam_of, local_of, descs, bounds):
        self.un_x = np.ascontiguousarray(un_x, np.float32); self.un_y = np.ascontiguousarray(un_y, np.float32)
        self.octave = np.ascontiguousarray(octave, np.int32); self.angle = np.ascontiguousarray(angle, np.float32)
        self.uright = np.ascontiguousarray(uright, np.float32)
        self.cam_of = np.ascontiguousarray(cam_of, np.int32); self.local_of = np.ascontiguousarray(local_of, np.int32)
        self.descs = [_u8(d) for d in descs]
        self.bounds = tuple(float(b) for b in bounds)  # minX, minY, maxX, maxY
        self.n_total = len(self.un_x); self.n_cams = len(self.descs)
        self._ptrs = (C.c_void_p * self.n_cams)(*[d.ctypes.data for d in self.descs])
        self.c = Frame(self.n_total, self.n_cams, self.un_x.ctypes.data, self.un_y.ctypes.data, self.octave.ctypes.data,
                       self.angle.ctypes.data, self.uright.ctypes.data, self.cam_of.ctypes.data,
                       self.local_of.ctypes.data, C.cast(self._ptrs, C.c_void_p), *self.bounds)

    def ptr(self):
        return C.byref(self.c)


def grid_csr(frame):
    cs = np.zeros(frame.n_cams * 64 * 48 + 1, np.int32); items = np.zeros(max(frame.n_total, 1), np.int32)
    lib().orc_grid_csr(frame.ptr(), _p(cs), _p(items))
    return cs, items[:cs[-1]]


def features_in_area(frame, cam, x, y, r, min_level=-1, max_level=-1):
    out = np.zeros(max(frame.n_total, 1), np.int32)
    n = lib().orc_features_in_area(frame.ptr(), cam, x, y, r, min_level, max_level, _p(out), len(out))
    return out[:n].copy()


def project_best(frame, queries, occupied=None, gate=0, inv_sigma2=None):
    queries = np.ascontiguousarray(queries, QUERY_DTYPE); nq = len(queries)
    bi, bd = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    sg = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    lib().orc_project_best(frame.ptr(), _p(queries), nq, None if occ is None else _p(occ), gate, None if sg is None else _p(sg), _p(bi), _p(bd))
    return bi[:nq], bd[:nq]


def search_by_projection_frames(frame, queries, th_high=100, check_ori=True, occupied=None):
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    m = np.zeros(max(frame.n_total, 1), np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    n = lib().orc_search_by_projection_frames(frame.ptr(), _p(queries), len(queries), None if occ is None else _p(occ),
                                              th_high, int(check_ori), _p(m))
    return n, m[:frame.n_total]


def search_by_projection_loop2(frame, queries, windows2, occupied=None, th_low=50):
    """Two-camera loop search (reference src/ORBmatcher.cc:566-750) from the projected windows on."""
    from multi_orb_slam_amd._lib import WINDOW_DTYPE
    queries = np.ascontiguousarray(queries, QUERY_DTYPE); windows2 = np.ascontiguousarray(windows2, WINDOW_DTYPE)
    assert len(queries) == len(windows2)
    m = np.zeros(max(frame.n_total, 1), np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    n = lib().orc_search_by_projection_loop2(frame.ptr(), _p(queries), _p(windows2), len(queries), None if occ is None else _p(occ),
                                             th_low, _p(m))
    return n, m[:frame.n_total]


def search_for_initialization(frame2, queries, nnratio=0.9, check_ori=True, th_low=50):
    """SearchForInitialization (reference src/ORBmatcher.cc:868-994) from the kept level-0 keypoints on -> (nmatches, match12)."""
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    m = np.zeros(max(len(queries), 1), np.int32)
    n = lib().orc_search_for_initialization(frame2.ptr(), _p(queries), len(queries), C.c_float(nnratio), int(check_ori), th_low, _p(m))
    return n, m[:len(queries)]


def search_by_projection_points(frame, queries, occupied=None, nnratio=0.8, th_high=100):
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    m = np.zeros(max(frame.n_total, 1), np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    n = lib().orc_search_by_projection_points(frame.ptr(), _p(queries), len(queries),
                                              None if occ is None else _p(occ), nnratio, th_high, _p(m))
    return n, m[:frame.n_total]


# ---------------------------------------------------------------------------------------------- vocabulary / BoW searches
class Vocabulary:
    def __init__(self, voc):
        self._keep = [np.ascontiguousarray(voc["parent"], np.int32), np.ascontiguousarray(voc["is_leaf"], np.uint8),
                      np.ascontiguousarray(voc["desc"], np.uint8), np.ascontiguousarray(voc["weight"], np.float64)]
        self._h = lib().orc_vocab_create(len(self._keep[0]), int(voc["L"]), *[_p(a) for a in self._keep])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_vocab_destroy(self._h); self._h = None

    def transform(self, features, levelsup=4):
        f = _u8(features).reshape(-1, 32); n = len(f)
        w, nd, wt = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float64)
        lib().orc_bow_transform(self._h, _p(f), n, levelsup, _p(w), _p(nd), _p(wt))
        return w, nd, wt

    def bow_vectors(self, features, levelsup=4):
        """-> ((word ids, values), (node ids, node_start, items))"""
        f = _u8(features).reshape(-1, 32); n = len(f)
        bid, bval = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fn, fs, fi = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.uint32)
        nn = C.c_int()
        nw = lib().orc_bow_vectors(self._h, _p(f), n, levelsup, _p(bid), _p(bval), _p(fn), _p(fs), _p(fi), C.byref(nn))
        return (bid[:nw].copy(), bval[:nw].copy()), (fn[:nn.value].copy(), fs[:nn.value + 1].copy(), fi[:fs[nn.value]].copy())


def bow_score_l1(a, b):
    ia, va = np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.float64)
    ib, vb = np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.float64)
    return lib().orc_bow_score_l1(_p(ia), _p(va), len(ia), _p(ib), _p(vb), len(ib))


class BowSideC(C.Structure):
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("angle", C.c_void_p), ("flags", C.c_void_p), ("n_nodes", C.c_int),
                ("node_id", C.c_void_p), ("node_start", C.c_void_p), ("items", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
                ("octave", C.c_void_p), ("cam_of", C.c_void_p)]


class TriangulationC(C.Structure):
    _fields_ = [("n_cams", C.c_int), ("n_levels", C.c_int), ("F12", C.c_void_p), ("ex", C.c_void_p), ("ey", C.c_void_p),
                ("scale_factors", C.c_void_p), ("level_sigma2", C.c_void_p)]


def _bow_side(s):
    """s: dict(desc, angle, flags, node_id, node_start, items[, x, y, octave, cam_of]) -> (struct, keepalive)"""
    keep = dict(desc=_u8(s["desc"]).reshape(-1, 32), angle=np.ascontiguousarray(s["angle"], np.float32),
                flags=np.ascontiguousarray(s["flags"], np.uint8), node_id=np.ascontiguousarray(s["node_id"], np.uint32),
                node_start=np.ascontiguousarray(s["node_start"], np.int32), items=np.ascontiguousarray(s["items"], np.uint32))
    for k, dt in (("x", np.float32), ("y", np.float32), ("octave", np.int32), ("cam_of", np.int32)):
        keep[k] = np.ascontiguousarray(s[k], dt) if s.get(k) is not None else None
    g = lambda k: keep[k].ctypes.data if keep[k] is not None else None
    st = BowSideC(len(keep["desc"]), g("desc"), g("angle"), g("flags"), len(keep["node_id"]), g("node_id"), g("node_start"), g("items"),
                  g("x"), g("y"), g("octave"), g("cam_of"))
    return st, keep


def search_by_bow(a, b, mode, th_low=50, nnratio=0.7, check_ori=True):
    sa, ka = _bow_side(a); sb, kb = _bow_side(b)
    n_out = sb.n if mode == 0 else sa.n
    match = np.full(max(n_out, 1), -1, np.int32)
    nm = lib().orc_search_by_bow(C.byref(sa), C.byref(sb), mode, th_low, nnratio, int(check_ori), _p(match))
    return nm, match[:n_out]


def search_for_triangulation(a, b, F12, ex, ey, scale_factors, level_sigma2, th_low=50, check_ori=True):
    sa, ka = _bow_side(a); sb, kb = _bow_side(b)
    F = np.ascontiguousarray(F12, np.float32).reshape(-1, 9)
    exa, eya = np.ascontiguousarray(ex, np.float32), np.ascontiguousarray(ey, np.float32)
    sf, s2 = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
    T = TriangulationC(len(F), len(sf), F.ctypes.data, exa.ctypes.data, eya.ctypes.data, sf.ctypes.data, s2.ctypes.data)
    match = np.full(max(sa.n, 1), -1, np.int32)
    nm = lib().orc_search_for_triangulation(C.byref(sa), C.byref(sb), C.byref(T), th_low, int(check_ori), _p(match))
    return nm, match[:sa.n]
