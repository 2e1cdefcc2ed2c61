"""Host-side mirror of ORB_SLAM2::ORBextractor over the C ABI (include/orbx.h).

Mirrors the reference class (include/ORBextractor.h:45-112): constructor arguments, `__call__(image)` ==
operator()(image, mask, keypoints, descriptors) for one camera, the six getters -- plus the N-camera batched entry
the MI355X design adds.  All compute happens in libmorb.so's HIP kernels.
"""
import ctypes as C
from dataclasses import dataclass
import numpy as np
from . import _lib
from ._lib import KP_DTYPE, Params, check, ptr


@dataclass
class ExtractorParams:
    nfeatures: int = 1000       # OtherFiles/multi.yaml:42-55 defaults
    scale_factor: float = 1.2
    nlevels: int = 8
    ini_th_fast: int = 20
    min_th_fast: int = 7

    def c(self):
        return Params(self.nfeatures, self.scale_factor, self.nlevels, self.ini_th_fast, self.min_th_fast)


def tables(p: ExtractorParams):
    """Scale / sigma tables, per-level quotas and umax (reference src/ORBextractor.cc:416-470)."""
    n = p.nlevels
    sc = np.zeros(n, np.float32); inv = np.zeros(n, np.float32); s2 = np.zeros(n, np.float32); is2 = np.zeros(n, np.float32)
    quota = np.zeros(n, np.int32); umax = np.zeros(16, np.int32)
    cp = p.c()
    check(_lib.lib().orbx_tables(C.byref(cp), ptr(sc), ptr(inv), ptr(s2), ptr(is2), ptr(quota), ptr(umax)))
    return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)


def distribute_octree(kps, min_x, max_x, min_y, max_y, n_features):
    """The library's host quadtree on caller-supplied candidates (no GPU needed)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros(max(len(kps), 1), KP_DTYPE); n = C.c_int()
    check(_lib.lib().orbx_debug_distribute_octree(ptr(kps), len(kps), min_x, max_x, min_y, max_y, n_features, ptr(out),
                                                  len(out), C.byref(n)))
    return out[:n.value]


class Extractor:
    """N-camera ORB extractor handle (one `orbx_extractor`)."""

    def __init__(self, params, max_width, max_height, device=0):
        if isinstance(params, ExtractorParams):
            params = [params]
        self.params = list(params)
        self.n_cams = len(self.params)
        arr = (Params * self.n_cams)(*[p.c() for p in self.params])
        self._h = C.c_void_p()
        check(_lib.lib().orbx_create(arr, self.n_cams, max_width, max_height, device, C.byref(self._h)))
        self._tables = [tables(p) for p in self.params]
        self.caps = [p.nfeatures + 4 * p.nlevels for p in self.params]

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().orbx_destroy(self._h)
            except Exception:
                pass
            self._h = None

    __del__ = close

    # -- reference getters (include/ORBextractor.h:64-84), camera 0 unless told otherwise
    def GetLevels(self, cam=0): return self.params[cam].nlevels
    def GetScaleFactor(self, cam=0): return self.params[cam].scale_factor
    def GetScaleFactors(self, cam=0): return self._tables[cam]["scale"]
    def GetInverseScaleFactors(self, cam=0): return self._tables[cam]["inv_scale"]
    def GetScaleSigmaSquares(self, cam=0): return self._tables[cam]["sigma2"]
    def GetInverseScaleSigmaSquares(self, cam=0): return self._tables[cam]["inv_sigma2"]

    @property
    def stream(self):
        return _lib.lib().orbx_stream(self._h)

    def extract(self, images):
        """== operator() for every camera: list of HxW uint8 arrays (None = empty image) -> [(keypoints, descriptors)]."""
        assert len(images) == self.n_cams
        L = _lib.lib()
        imgs = [None if im is None else np.ascontiguousarray(im, np.uint8) for im in images]
        gray = (C.c_void_p * self.n_cams)(*[None if im is None else im.ctypes.data for im in imgs])
        w = (C.c_int * self.n_cams)(*[0 if im is None else im.shape[1] for im in imgs])
        h = (C.c_int * self.n_cams)(*[0 if im is None else im.shape[0] for im in imgs])
        st = (C.c_int * self.n_cams)(*[0 if im is None else im.strides[0] for im in imgs])
        kps = [np.zeros(c, KP_DTYPE) for c in self.caps]
        desc = [np.zeros((c, 32), np.uint8) for c in self.caps]
        kp_p = (C.c_void_p * self.n_cams)(*[k.ctypes.data for k in kps])
        d_p = (C.c_void_p * self.n_cams)(*[d.ctypes.data for d in desc])
        cap = (C.c_int * self.n_cams)(*self.caps)
        n = (C.c_int * self.n_cams)()
        check(L.orbx_extract(self._h, self.n_cams, gray, w, h, st, kp_p, d_p, cap, n))
        return [(kps[c][:n[c]].copy(), desc[c][:n[c]].copy()) for c in range(self.n_cams)]

    def __call__(self, image, mask=None):
        """Single-camera operator(): returns (keypoints, descriptors); `mask` is ignored like in the reference."""
        assert self.n_cams == 1
        return self.extract([image])[0]

    # -- resident path
    def upload(self, cam, image):
        image = np.ascontiguousarray(image, np.uint8)
        check(_lib.lib().orbx_upload(self._h, cam, ptr(image), image.shape[1], image.shape[0], image.strides[0]))

    def upload_device(self, cam, dptr, width, height, stride):
        check(_lib.lib().orbx_upload_device(self._h, cam, C.c_void_p(dptr), width, height, stride))

    def run(self):
        check(_lib.lib().orbx_run(self._h))

    def count(self, cam):
        return _lib.lib().orbx_count(self._h, cam)

    def download(self, cam):
        n = self.count(cam)
        kps = np.zeros(max(n, 1), KP_DTYPE); desc = np.zeros((max(n, 1), 32), np.uint8)
        check(_lib.lib().orbx_download(self._h, cam, ptr(kps), ptr(desc), max(n, 1)))
        return kps[:n], desc[:n]

    def device_descriptors(self, cam):
        return _lib.lib().orbx_device_descriptors(self._h, cam)

    def device_keypoints(self, cam):
        return _lib.lib().orbx_device_keypoints(self._h, cam)

    def bind_output(self, cam, d_kps, d_desc, cap):
        check(_lib.lib().orbx_bind_output(self._h, cam, C.c_void_p(d_kps), C.c_void_p(d_desc), cap))

    def set_profiling(self, on=True):
        check(_lib.lib().orbx_set_profiling(self._h, int(on)))

    def stage_times_us(self):
        out = np.zeros(6, np.float32)
        check(_lib.lib().orbx_stage_times_us(self._h, ptr(out)))
        # "quadtree": DistributeOctTree wherever the last run did it -- the device kernel (k_octree; GPU time between the FAST and
        # describe stage events) or, on the host fallback, D2H + the host quadtree (wall time); "compact" is only non-trivial there
        return dict(zip(["pyramid", "fast_cells", "compact", "quadtree", "describe", "total_wall"], out.tolist()))

    # -- stage inspection
    def debug_level(self, cam, level):
        buf = np.zeros(1 << 24, np.uint8)
        w = C.c_int(); h = C.c_int()
        check(_lib.lib().orbx_debug_level(self._h, cam, level, ptr(buf), buf.size, C.byref(w), C.byref(h)))
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    def last_path(self):
        """0 device quadtree, 1 device quadtree incl. the memory-backed pass, 2 host quadtree (orbx_debug_last_path)."""
        return _lib.lib().orbx_debug_last_path(self._h)

    def pyramid_form(self):
        """0 k_pyramid_tiled, 2 k_resize per level (generic chain), 3 k_pyramid_tiled4 x 2 (orbx_debug_pyramid_form)"""
        return _lib.lib().orbx_debug_pyramid_form(self._h)

    def level0_in_place(self):
        """cameras whose level 0 the last run read in the caller's device buffer (orbx_debug_level0_in_place)"""
        return _lib.lib().orbx_debug_level0_in_place(self._h)

    def debug_candidates(self, cam, level):
        n = C.c_int()
        check(_lib.lib().orbx_debug_candidates(self._h, cam, level, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        check(_lib.lib().orbx_debug_candidates(self._h, cam, level, ptr(out), len(out), C.byref(n)))
        return out[:n.value]
