"""ctypes mirror of include/orbv.h: vocabulary-tree transform (DBoW2) and the BoW-gated searches of ORBmatcher."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import BowSide, Triangulation, check, ptr


class FeatureVector:
    """DBoW2::FeatureVector as CSR arrays: node ids ascending, feature indices per node in push_back order."""

    def __init__(self, node_id, node_start, items):
        self.node_id = np.ascontiguousarray(node_id, np.uint32)
        self.node_start = np.ascontiguousarray(node_start, np.int32)
        self.items = np.ascontiguousarray(items, np.uint32)

    def as_dict(self):
        return {int(k): self.items[self.node_start[i]:self.node_start[i + 1]].tolist() for i, k in enumerate(self.node_id)}


class Vocabulary:
    def __init__(self, parent=None, is_leaf=None, desc=None, weight=None, L=None, device=0, path=None):
        self._h = C.c_void_p()
        if path is not None:
            check(_lib.lib().orbv_load_text(str(path).encode(), device, C.byref(self._h)))
        else:
            parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
            check(_lib.lib().orbv_create(len(parent), int(L), ptr(parent), ptr(is_leaf), ptr(desc), ptr(weight), device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().orbv_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def stream(self):
        return _lib.lib().orbv_stream(self._h)

    def info(self):
        v = [C.c_int() for _ in range(4)]
        check(_lib.lib().orbv_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("n_nodes", "n_words", "k", "L"), (x.value for x in v)))

    def transform(self, features, levelsup=4):
        """-> (word_id, node_id, weight) per feature."""
        f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(f)
        w, nd, wt = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float64)
        check(_lib.lib().orbv_transform(self._h, ptr(f), n, levelsup, ptr(w), ptr(nd), ptr(wt)))
        return w, nd, wt

    def transform_device(self, d_features, n, levelsup, d_word, d_node, stream):
        check(_lib.lib().orbv_transform_device(self._h, C.c_void_p(d_features), n, levelsup, C.c_void_p(d_word), C.c_void_p(d_node),
                                               C.c_void_p(stream) if stream else None))

    def bow_vectors(self, features, levelsup=4):
        """transform(features, BowVector, FeatureVector, levelsup) -> ((word ids, values), FeatureVector)."""
        f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(f)
        bid, bval = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fn, fs, fi = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int(), C.c_int()
        check(_lib.lib().orbv_bow_vectors(self._h, ptr(f), n, levelsup, ptr(bid), ptr(bval), C.byref(nw), ptr(fn), ptr(fs), ptr(fi), C.byref(nn)))
        return (bid[:nw.value].copy(), bval[:nw.value].copy()), FeatureVector(fn[:nn.value], fs[:nn.value + 1], fi[:fs[nn.value]])


def score_l1(a, b):
    """L1Scoring::score of two (ids, values) sparse vectors."""
    ia, va = np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.float64)
    ib, vb = np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.float64)
    return _lib.lib().orbv_score_l1(ptr(ia), ptr(va), len(ia), ptr(ib), ptr(vb), len(ib))


class Side:
    """One frame / keyframe for the BoW searches (orbv_side); keeps the arrays alive."""

    def __init__(self, desc, angle, fv, flags=None, x=None, y=None, octave=None, cam_of=None):
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.n = len(self.desc)
        self.angle = np.ascontiguousarray(angle, np.float32)
        self.fv = fv
        self.flags = None if flags is None else np.ascontiguousarray(flags, np.uint8)
        self.x = None if x is None else np.ascontiguousarray(x, np.float32)
        self.y = None if y is None else np.ascontiguousarray(y, np.float32)
        self.octave = None if octave is None else np.ascontiguousarray(octave, np.int32)
        self.cam_of = None if cam_of is None else np.ascontiguousarray(cam_of, np.int32)

    def c(self):
        p = lambda a: None if a is None else a.ctypes.data
        return BowSide(self.n, p(self.desc), p(self.angle), p(self.flags), len(self.fv.node_id), p(self.fv.node_id), p(self.fv.node_start),
                       p(self.fv.items), p(self.x), p(self.y), p(self.octave), p(self.cam_of))


class BowSearch:
    """Stream + scratch of the BoW-gated searches (orbv_workspace)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(_lib.lib().orbv_workspace_create(device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().orbv_workspace_destroy(self._h)
            self._h = None

    __del__ = close

    def search_by_bow(self, a, b, mode, th_low=50, nnratio=0.7, check_orientation=True):
        """mode 0: SearchByBoW(KF a, Frame b) -> match per feature of b; mode 1: (KF a, KF b) -> match per feature of a."""
        n_out = b.n if mode == 0 else a.n
        match = np.full(max(n_out, 1), -1, np.int32); nm = C.c_int()
        ca, cb = a.c(), b.c()
        check(_lib.lib().orbv_search_by_bow(self._h, C.byref(ca), C.byref(cb), mode, th_low, nnratio, int(check_orientation), ptr(match), C.byref(nm)))
        return nm.value, match[:n_out]

    def keyframe_from_device(self, vocabulary, feats, levelsup=4, triangulation=True):
        """Keyframe built entirely on the device from a front end's resident frame (feats = NativeFrontEnd.export_features())."""
        from ._lib import DeviceSide
        s = DeviceSide()
        s.n = feats.n_total; s.d_desc = feats.d_desc; s.d_angle = feats.d_angle; s.d_uright = feats.d_uright
        if triangulation:
            s.d_x = feats.d_un_x; s.d_y = feats.d_un_y; s.d_octave = feats.d_octave
        s.n_cams = feats.n_cams
        acc = 0
        for c in range(feats.n_cams):
            s.cam_start[c] = acc; acc += feats.counts[c]
        s.cam_start[feats.n_cams] = acc
        k = Keyframe.__new__(Keyframe)
        k._h = C.c_void_p(); k.n = feats.n_total; k._search = self
        check(_lib.lib().orbv_keyframe_from_device(self._h, vocabulary._h, C.byref(s), levelsup, C.c_void_p(feats.stream) if feats.stream else None, C.byref(k._h)))
        return k

    def keyframe_download(self, k):
        """-> (word_id, node_of_feature, FeatureVector) of a device-built keyframe."""
        n = k.n
        w, nd = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        fn, fs, fi = np.zeros(4100, np.uint32), np.zeros(4101, np.int32), np.zeros(max(n, 1), np.uint32)
        nn = C.c_int()
        check(_lib.lib().orbv_keyframe_download(self._h, k._h, ptr(w), ptr(nd), ptr(fn), ptr(fs), ptr(fi), C.byref(nn)))
        return w[:n], nd[:n], FeatureVector(fn[:nn.value], fs[:nn.value + 1], fi[:fs[nn.value]])

    def keyframe(self, side):
        """Upload a Side once (orbv_keyframe_create); pass the result wherever a Side is accepted."""
        return Keyframe(self, side)

    @staticmethod
    def _flags(f):
        return None if f is None else np.ascontiguousarray(f, np.uint8)

    def search_by_bow_resident(self, a, b, mode, flags_a=None, flags_b=None, th_low=50, nnratio=0.7, check_orientation=True):
        n_out = b.n if mode == 0 else a.n
        match = np.full(max(n_out, 1), -1, np.int32); nm = C.c_int()
        fa, fb = self._flags(flags_a), self._flags(flags_b)
        check(_lib.lib().orbv_search_by_bow_resident(self._h, a._h, None if fa is None else ptr(fa), b._h, None if fb is None else ptr(fb), mode,
                                                     th_low, nnratio, int(check_orientation), ptr(match), C.byref(nm)))
        return nm.value, match[:n_out]

    def search_for_triangulation_resident(self, a, b, F12, ex, ey, scale_factors, level_sigma2, flags_a=None, flags_b=None, th_low=50,
                                          check_orientation=True):
        T, keep = self._tri(F12, ex, ey, scale_factors, level_sigma2)
        match = np.full(max(a.n, 1), -1, np.int32); nm = C.c_int()
        fa, fb = self._flags(flags_a), self._flags(flags_b)
        check(_lib.lib().orbv_search_for_triangulation_resident(self._h, a._h, None if fa is None else ptr(fa), b._h, None if fb is None else ptr(fb),
                                                                C.byref(T), th_low, int(check_orientation), ptr(match), C.byref(nm)))
        return nm.value, match[:a.n]

    @staticmethod
    def _tri(F12, ex, ey, scale_factors, level_sigma2):
        T = Triangulation()
        F12 = np.ascontiguousarray(F12, np.float32).reshape(-1, 9)
        T.n_cams = len(F12)
        sf = np.ascontiguousarray(scale_factors, np.float32); s2 = np.ascontiguousarray(level_sigma2, np.float32)
        T.n_levels = len(sf)
        for c in range(len(F12)):
            for k in range(9):
                T.F12[c][k] = float(F12[c, k])
            T.ex[c] = float(ex[c]); T.ey[c] = float(ey[c])
        T.scale_factors = sf.ctypes.data; T.level_sigma2 = s2.ctypes.data
        return T, (sf, s2)

    def search_for_triangulation(self, a, b, F12, ex, ey, scale_factors, level_sigma2, th_low=50, check_orientation=True):
        T = Triangulation()
        F12 = np.ascontiguousarray(F12, np.float32).reshape(-1, 9)
        T.n_cams = len(F12)
        sf = np.ascontiguousarray(scale_factors, np.float32); s2 = np.ascontiguousarray(level_sigma2, np.float32)
        T.n_levels = len(sf)
        for c in range(len(F12)):
            for k in range(9):
                T.F12[c][k] = float(F12[c, k])
            T.ex[c] = float(ex[c]); T.ey[c] = float(ey[c])
        T.scale_factors = sf.ctypes.data; T.level_sigma2 = s2.ctypes.data
        match = np.full(max(a.n, 1), -1, np.int32); nm = C.c_int()
        ca, cb = a.c(), b.c()
        check(_lib.lib().orbv_search_for_triangulation(self._h, C.byref(ca), C.byref(cb), C.byref(T), th_low, int(check_orientation), ptr(match), C.byref(nm)))
        return nm.value, match[:a.n]


class Keyframe:
    """A Side resident in HBM (orbv_keyframe)."""

    def __init__(self, search, side):
        self._h = C.c_void_p(); self.n = side.n; self._search = search
        cs = side.c()
        check(_lib.lib().orbv_keyframe_create(search._h, C.byref(cs), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().orbv_keyframe_destroy(self._h)
            self._h = None

    __del__ = close
