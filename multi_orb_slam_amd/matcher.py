"""Host-side mirror of the hot part of ORB_SLAM2::ORBmatcher over the C ABI (include/orbm.h)."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import KP_DTYPE, QUERY_DTYPE, CamFeatures, FrameDesc, check, ptr

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # reference src/ORBmatcher.cc:37-39


def descriptor_distance(a, b):
    """static ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:3994-4010)."""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return _lib.lib().orbm_descriptor_distance(ptr(a), ptr(b))


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32); ind = np.zeros(3, np.int32)
    _lib.lib().orbm_three_maxima(ptr(sizes), len(sizes), ptr(ind))
    return tuple(int(i) for i in ind)


class FrameData:
    """Flat arrays of the Frame members the matcher reads (reference src/Frame.cc:191-288): cam-major global index."""

    def __init__(self, un_x, un_y, octave, angle, uright, cam_of, local_of, descs, bounds):
        self.un_x = np.ascontiguousarray(un_x, np.float32); self.un_y = np.ascontiguousarray(un_y, np.float32)
        self.octave = np.ascontiguousarray(octave, np.int32); self.angle = np.ascontiguousarray(angle, np.float32)
        self.uright = np.ascontiguousarray(uright, np.float32)
        self.cam_of = np.ascontiguousarray(cam_of, np.int32); self.local_of = np.ascontiguousarray(local_of, np.int32)
        self.descs = [np.ascontiguousarray(d, np.uint8) for d in descs]
        self.bounds = tuple(float(b) for b in bounds)
        self.n_total = len(self.un_x); self.n_cams = len(self.descs)
        self._ptrs = (C.c_void_p * self.n_cams)(*[d.ctypes.data for d in self.descs])
        self.c = FrameDesc(self.n_total, self.n_cams, self.un_x.ctypes.data, self.un_y.ctypes.data,
                           self.octave.ctypes.data, self.angle.ctypes.data, self.uright.ctypes.data,
                           self.cam_of.ctypes.data, self.local_of.ctypes.data, C.cast(self._ptrs, C.c_void_p),
                           *self.bounds)

    @staticmethod
    def from_cameras(per_cam, width, height, uright=None):
        """Frame merge (reference src/Frame.cc:191-239): per_cam = [(keypoints, descriptors), ...] -> `_total` arrays.
        Undistortion is the identity (k1 == 0) and the image bounds are [0,W]x[0,H] (Frame.cc:743-779)."""
        xs, ys, octs, angs, cams, locs, descs = [], [], [], [], [], [], []
        for c, (k, d) in enumerate(per_cam):
            xs.append(k["x"]); ys.append(k["y"]); octs.append(k["octave"]); angs.append(k["angle"])
            cams.append(np.full(len(k), c, np.int32)); locs.append(np.arange(len(k), dtype=np.int32)); descs.append(d)
        cat = np.concatenate
        n = sum(len(x) for x in xs)
        ur = np.full(n, -1.0, np.float32) if uright is None else uright
        return FrameData(cat(xs), cat(ys), cat(octs), cat(angs), ur, cat(cams), cat(locs), descs, (0, 0, width, height))


class _Count:
    def __init__(self, n_total, n_cams):
        self.n_total, self.n_cams = n_total, n_cams


class Frame:
    def __init__(self, matcher, data=None, handle=None, n_total=0, n_cams=0, resident=None):
        """resident: None -> orbm_frame_create (host-built grid, every array sent); a list with one entry per camera -- a DEVICE
        pointer to that camera's descriptor rows, or 0 / None -- -> orbm_frame_create_resident (grid built on the device, the
        named cameras' descriptors read where they are)."""
        self._m = matcher
        if handle is not None:           # device-built frame (orbm_frame_from_device)
            self._h = handle
            self.data = _Count(n_total, n_cams)
        elif resident is not None:
            self.data = data
            self._h = C.c_void_p()
            ptrs = (C.c_void_p * data.n_cams)(*[(p or None) for p in resident])
            check(_lib.lib().orbm_frame_create_resident(matcher._h, C.byref(data.c), ptrs, C.byref(self._h)))
        else:
            self.data = data
            self._h = C.c_void_p()
            check(_lib.lib().orbm_frame_create(matcher._h, C.byref(data.c), C.byref(self._h)))

    def download(self, kps=True, desc=True, uright=True, depth=True):
        """Host copies of the merged arrays of a device-built frame (global, cam-major order)."""
        n = max(self.data.n_total, 1)
        k = np.zeros(n, KP_DTYPE) if kps else None
        d = np.zeros((n, 32), np.uint8) if desc else None
        ur = np.zeros(n, np.float32) if uright else None
        dp = np.zeros(n, np.float32) if depth else None
        check(_lib.lib().orbm_frame_download(self._m._h, self._h, None if k is None else ptr(k), None if d is None else ptr(d),
                                             None if ur is None else ptr(ur), None if dp is None else ptr(dp)))
        nt = self.data.n_total
        return tuple(None if a is None else a[:nt] for a in (k, d, ur, dp))

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().orbm_frame_destroy(self._h)
            except Exception:
                pass
            self._h = None

    __del__ = close

    def grid(self):
        cs = np.zeros(self.data.n_cams * 64 * 48 + 1, np.int32); items = np.zeros(max(self.data.n_total, 1), np.int32)
        check(_lib.lib().orbm_frame_grid(self._h, ptr(cs), ptr(items)))
        return cs, items[:cs[-1]]


class Matcher:
    """ORBmatcher(nnratio=0.6, checkOri=True) (reference include/ORBmatcher.h:41)."""

    def __init__(self, nnratio=0.6, check_orientation=True, device=0):
        self.nnratio = float(nnratio); self.check_orientation = bool(check_orientation)
        self._h = C.c_void_p()
        check(_lib.lib().orbm_create(device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().orbm_destroy(self._h)
            except Exception:  # interpreter teardown
                pass
            self._h = None

    __del__ = close

    @property
    def stream(self):
        return _lib.lib().orbm_stream(self._h)

    DescriptorDistance = staticmethod(descriptor_distance)

    def hamming_top2(self, q, r):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); r = np.ascontiguousarray(r, np.uint8).reshape(-1, 32)
        bi = np.zeros(len(q), np.int32); bd = np.zeros(len(q), np.int32); sd = np.zeros(len(q), np.int32)
        check(_lib.lib().orbm_hamming_top2(self._h, ptr(q), len(q), ptr(r), len(r), ptr(bi), ptr(bd), ptr(sd)))
        return bi, bd, sd

    def hamming_matrix(self, q, r):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); r = np.ascontiguousarray(r, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), len(r)), np.uint16)
        check(_lib.lib().orbm_hamming_matrix(self._h, ptr(q), len(q), ptr(r), len(r), ptr(out)))
        return out

    # -- device-pointer entry points (asynchronous on `stream`)
    @staticmethod
    def top2_scratch_bytes(nq, nr):
        return int(_lib.lib().orbm_top2_scratch_bytes(nq, nr))

    @staticmethod
    def hamming_top2_device(d_q, nq, d_r, nr, d_best_idx, d_best_dist, d_second, d_scratch, stream):
        check(_lib.lib().orbm_hamming_top2_device(C.c_void_p(d_q), nq, C.c_void_p(d_r), nr, C.c_void_p(d_best_idx),
                                                  C.c_void_p(d_best_dist), C.c_void_p(d_second), C.c_void_p(d_scratch),
                                                  C.c_void_p(stream)))

    @staticmethod
    def use_matrix_cores(on):
        """orbm_use_matrix_cores: 1 / 0 = matrix-core / popcount form of the all-pairs kernels, -1 = default; returns the
        previous setting."""
        return int(_lib.lib().orbm_use_matrix_cores(int(on)))

    @staticmethod
    def use_fp4_top2(on):
        """orbm_use_fp4_top2: 1 / 0 = FP4 / int8 arithmetic in the matrix-core top-2 kernels, -1 = default; returns the previous
        setting."""
        return int(_lib.lib().orbm_use_fp4_top2(int(on)))

    @staticmethod
    def hamming_matrix_device(d_q, nq, d_r, nr, d_out, stream):
        check(_lib.lib().orbm_hamming_matrix_device(C.c_void_p(d_q), nq, C.c_void_p(d_r), nr, C.c_void_p(d_out),
                                                    C.c_void_p(stream)))

    def frame(self, data, resident=None):
        return Frame(self, data, resident=resident)

    def set_stream(self, stream):
        check(_lib.lib().orbm_set_stream(self._h, C.c_void_p(stream) if stream else None))

    def frame_from_device(self, cams, mbf, bounds):
        """Frame assembly on the device (merge + depth -> uRight + grid) from HBM-resident per-camera outputs.
        cams: [(d_kps, d_desc, n, d_depth or 0, depth_stride)]; bounds = (min_x, min_y, max_x, max_y)."""
        arr = (CamFeatures * len(cams))(*[CamFeatures(c[0], c[1], c[2], c[3] or None, c[4]) for c in cams])
        h = C.c_void_p()
        check(_lib.lib().orbm_frame_from_device(self._h, arr, len(cams), mbf, bounds[0], bounds[1], bounds[2], bounds[3],
                                                C.byref(h)))
        return Frame(self, handle=h, n_total=sum(c[2] for c in cams), n_cams=len(cams))

    def cross_top2_blocks(self, block_ptrs, counts, first_query_block, n_query_blocks):
        """Cross-camera top-2 over HBM-resident descriptor blocks (one per camera of the whole rig)."""
        nb = len(block_ptrs)
        ptrs = (C.c_void_p * nb)(*block_ptrs)
        cnt = (C.c_int * nb)(*counts)
        nq = sum(counts[first_query_block:first_query_block + n_query_blocks])
        bi = np.zeros(max(nq, 1), np.int32); bd = np.zeros(max(nq, 1), np.int32); sd = np.zeros(max(nq, 1), np.int32)
        check(_lib.lib().orbm_cross_top2_blocks(self._h, ptrs, cnt, nb, first_query_block, n_query_blocks, ptr(bi), ptr(bd),
                                                ptr(sd)))
        return bi[:nq], bd[:nq], sd[:nq]

    def wait_for_stream(self, stream_handle):
        """Order this matcher's stream behind another HIP stream's work so far (orbm_wait_for_stream); no host wait."""
        check(_lib.lib().orbm_wait_for_stream(self._h, C.c_void_p(stream_handle)))

    def last_resolve(self):
        """(status, matches, sweeps, longest candidate list) of the last device resolve (orbm_debug_last_resolve)."""
        out = (C.c_int * 4)()
        check(_lib.lib().orbm_debug_last_resolve(self._h, out))
        return tuple(out)

    def cross_top2_gathered(self, gathered_ptr, world, block_bytes, cap_rows, cams_per_rank, rank):
        """Cross-camera top-2 of this rank's features against the whole rig from ONE all-gathered buffer
        (orbm_cross_top2_gathered).  -> (best_idx, best_dist, second_dist, counts of every camera of the rig)."""
        key = (cap_rows, world * cams_per_rank)
        buf = getattr(self, "_gathered_buf", None)
        if buf is None or buf[0] != key:      # result buffers are reused from call to call (copied out below)
            buf = self._gathered_buf = (key, np.zeros(cap_rows, np.int32), np.zeros(cap_rows, np.int32), np.zeros(cap_rows, np.int32),
                                        np.zeros(world * cams_per_rank, np.int32))
        _, bi, bd, sd, cnt = buf
        nq = C.c_int()
        check(_lib.lib().orbm_cross_top2_gathered(self._h, C.c_void_p(gathered_ptr), world, block_bytes, cap_rows, cams_per_rank, rank,
                                                  ptr(bi), ptr(bd), ptr(sd), ptr(cnt), C.byref(nq)))
        return bi[:nq.value].copy(), bd[:nq.value].copy(), sd[:nq.value].copy(), cnt.tolist()

    def cross_top2_gathered_enqueue(self, gathered_ptr, world, block_bytes, cap_rows, cams_per_rank, rank, after_stream=None):
        """Enqueue half of cross_top2_gathered (side stream, joined into the main stream).  after_stream: raw handle of the
        stream the gathered buffer is produced on (0 = the default stream); None: no ordering needed."""
        self._gathered_shape = (cap_rows, world * cams_per_rank)
        check(_lib.lib().orbm_cross_top2_gathered_enqueue(self._h, C.c_void_p(gathered_ptr), world, block_bytes, cap_rows, cams_per_rank,
                                                          rank, C.c_void_p(after_stream or 0), 0 if after_stream is None else 1))

    def cross_top2_gathered_collect_views(self):
        """Collect half without copies: (best_idx, best_dist, second_dist) as views of the native pinned arrays + counts."""
        from .frontend import _view
        cap_rows, n_cams = self._gathered_shape
        cnt = np.zeros(n_cams, np.int32); nq = C.c_int()
        check(_lib.lib().orbm_cross_top2_gathered_collect(self._h, None, None, None, ptr(cnt), C.byref(nq)))
        p = [C.c_void_p() for _ in range(3)]
        check(_lib.lib().orbm_cross_top2_gathered_views(self._h, *[C.byref(x) for x in p]))
        n = nq.value
        return tuple(_view(x.value, np.int32, n) for x in p) + (cnt.tolist(),)

    def cross_top2_gathered_collect(self):
        """Collect half: after the handle's main stream has been synchronised (orbf_step_end)."""
        cap_rows, n_cams = self._gathered_shape
        bi = np.zeros(cap_rows, np.int32); bd = np.zeros(cap_rows, np.int32); sd = np.zeros(cap_rows, np.int32)
        cnt = np.zeros(n_cams, np.int32); nq = C.c_int()
        check(_lib.lib().orbm_cross_top2_gathered_collect(self._h, ptr(bi), ptr(bd), ptr(sd), ptr(cnt), C.byref(nq)))
        return bi[:nq.value], bd[:nq.value], sd[:nq.value], cnt.tolist()

    def cross_top2(self, frame):
        n = max(frame.data.n_total, 1)
        bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32); sd = np.zeros(n, np.int32)
        check(_lib.lib().orbm_cross_top2(self._h, frame._h, ptr(bi), ptr(bd), ptr(sd)))
        nt = frame.data.n_total
        return bi[:nt], bd[:nt], sd[:nt]

    def features_in_area(self, frame, cam, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(frame.data.n_total, 1), np.int32); n = C.c_int()
        check(_lib.lib().orbm_features_in_area(self._h, frame._h, cam, x, y, r, min_level, max_level, ptr(out), len(out),
                                               C.byref(n)))
        return out[:n.value].copy()

    def project_best(self, frame, queries, occupied=None, gate=0, inv_level_sigma2=None):
        """Nearest candidate of every projected point on its own (orbm_project_best: the inner loop of SearchBySim3 / Fuse).
        gate 0 none, 1 right-coordinate window, 2 Fuse's chi-square gate (needs inv_level_sigma2)."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE); nq = len(queries)
        bi, bd = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        check(_lib.lib().orbm_project_best(self._h, frame._h, ptr(queries), nq, None if occ is None else ptr(occ), gate,
                                           None if sg is None else ptr(sg), 0 if sg is None else len(sg), ptr(bi), ptr(bd)))
        return bi[:nq], bd[:nq]

    def time_project(self, frame, queries, th_high=TH_HIGH, iters=50):
        """(average launch duration of the projection kernel in microseconds, candidates that passed the gates):
        orbm_debug_time_project -- the kernel alone, as the frame search launches it (roofline M3)."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        us = C.c_float(); n = C.c_longlong()
        check(_lib.lib().orbm_debug_time_project(self._h, frame._h, ptr(queries), len(queries), th_high, iters, C.byref(us), C.byref(n)))
        return us.value, n.value

    def project_candidates(self, frame, queries, cap):
        queries = np.ascontiguousarray(queries, QUERY_DTYPE); nq = len(queries)
        idx = np.zeros((max(nq, 1), cap), np.int32); dist = np.zeros((max(nq, 1), cap), np.uint16)
        cnt = np.zeros(max(nq, 1), np.int32)
        check(_lib.lib().orbm_project_candidates(self._h, frame._h, ptr(queries), nq, cap, ptr(idx), ptr(dist), ptr(cnt)))
        return idx[:nq], dist[:nq], cnt[:nq]

    def SearchByProjection(self, frame, queries, th_high=TH_HIGH, occupied=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono, Calib) from the projected queries on
        (reference src/ORBmatcher.cc:3448-3641).  Returns (nmatches, match_of_feature); match_of_feature[g] is the
        query index, -1 (untouched) or -2 (cleared by the rotation-histogram filter)."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        m = np.zeros(max(frame.data.n_total, 1), np.int32); n = C.c_int()
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        check(_lib.lib().orbm_search_by_projection(self._h, frame._h, ptr(queries), len(queries),
                                                   None if occ is None else ptr(occ), th_high,
                                                   int(self.check_orientation), ptr(m), C.byref(n)))
        return n.value, m[:frame.data.n_total]

    def SearchByProjectionWindows(self, frame, queries, windows2, th_high=TH_LOW, occupied=None):
        """Two-camera loop search (reference src/ORBmatcher.cc:566-750) from the projected windows on: queries[i] carries the
        camera-1 window, windows2[i] the camera-2 window (cam < 0: none); orbm_search_by_projection_windows."""
        from ._lib import WINDOW_DTYPE
        queries = np.ascontiguousarray(queries, QUERY_DTYPE); windows2 = np.ascontiguousarray(windows2, WINDOW_DTYPE)
        assert len(queries) == len(windows2)
        m = np.zeros(max(frame.data.n_total, 1), np.int32); n = C.c_int()
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        check(_lib.lib().orbm_search_by_projection_windows(self._h, frame._h, ptr(queries), ptr(windows2), len(queries),
                                                           None if occ is None else ptr(occ), th_high, 0, ptr(m), C.byref(n)))
        return n.value, m[:frame.data.n_total]

    def SearchByProjectionPoints(self, frame, queries, occupied=None, th_high=TH_HIGH):
        """SearchByProjection(F, vpMapPoints, th) (reference src/ORBmatcher.cc:62-149)."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        m = np.zeros(max(frame.data.n_total, 1), np.int32); n = C.c_int()
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        check(_lib.lib().orbm_search_by_projection_points(self._h, frame._h, ptr(queries), len(queries),
                                                          None if occ is None else ptr(occ), self.nnratio, th_high,
                                                          ptr(m), C.byref(n)))
        return n.value, m[:frame.data.n_total]
