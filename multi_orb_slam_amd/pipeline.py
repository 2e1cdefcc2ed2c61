"""One front-end timestep over the C ABI: N-camera extract -> frame merge -> temporal SearchByProjection ->
cross-camera exhaustive top-2.  This is what bench.py times, what smoke() runs once, and what each rank of the
multi-GPU driver runs on its own cameras (the cross-camera step then matches against every rank's descriptors,
exchanged with one all-gather).

The step mirrors the reference's per-frame front end: Frame::Frame (src/Frame.cc:148-288: two extractor calls + the
`_total` merge + grid), TrackWithMotionModel's SearchByProjection (src/Tracking.cc:1267 -> src/ORBmatcher.cc:3448) and
the brute-force top-2 inner loop with SearchByBoW's acceptance rule (src/ORBmatcher.cc:287-327).  The projection of
last-frame points is host arithmetic in the reference too; here the synthetic stream's known motion plays the
motion model.
"""
import numpy as np
from ._lib import KP_DTYPE, QUERY_DTYPE
from .extractor import Extractor
from .matcher import Matcher, FrameData, TH_LOW
from . import rt

MOTION = (3.0, 1.0)      # synthetic stream: frame t+1 = frame t translated by (3, 1) px
TH_PROJ = 15.0           # TrackWithMotionModel search radius parameter (src/Tracking.cc:1267)
BOW_RATIO = 0.7          # ORBmatcher(0.7, true).SearchByBoW (src/Tracking.cc:1076)
BF = 40.0                # synthetic stereo baseline * fx


def synth_depth(x, y):
    """Deterministic stand-in for the depth image lookup (src/Frame.cc:959-1034): depth from the pixel position."""
    xi = x.astype(np.int64); yi = y.astype(np.int64)
    return (2.0 + ((xi * 31 + yi * 17) % 64) / 8.0).astype(np.float32)


def synth_uright(kps):
    d = synth_depth(kps["x"], kps["y"])
    return (kps["x"] - np.float32(BF) / d).astype(np.float32)


def make_queries(prev_per_cam, scale_factors, cam_offset=0):
    """Last-frame map points projected into the current frame (what src/ORBmatcher.cc:3502-3552 computes on the host)."""
    qs = []
    for c, (k, d) in enumerate(prev_per_cam):
        q = np.zeros(len(k), QUERY_DTYPE)
        q["u"] = k["x"] + np.float32(MOTION[0]); q["v"] = k["y"] + np.float32(MOTION[1])
        q["radius"] = np.float32(TH_PROJ) * scale_factors[k["octave"]]
        q["ur"] = q["u"] - np.float32(BF) / synth_depth(k["x"], k["y"])
        q["min_level"] = k["octave"] - 1; q["max_level"] = k["octave"] + 1
        q["cam"] = c + cam_offset; q["blocks"] = 1; q["angle"] = k["angle"]; q["desc"] = d
        qs.append(q)
    return np.concatenate(qs) if qs else np.zeros(0, QUERY_DTYPE)


def accept_cross(best_dist, second_dist):
    """SearchByBoW acceptance (src/ORBmatcher.cc:324-327): best <= TH_LOW and best < ratio * second (float compare)."""
    return (best_dist <= TH_LOW) & (best_dist.astype(np.float32) < np.float32(BOW_RATIO) * second_dist.astype(np.float32))


class FrontEnd:
    """Extractor + matcher for the cameras owned by this process (one process per GPU)."""

    def __init__(self, params_per_cam, width, height, device=0, rank=0, world_size=1, gather=None):
        self.params = list(params_per_cam); self.n_cams = len(self.params)
        self.width, self.height = width, height
        self.rank, self.world = rank, world_size
        self.gather = gather                      # callable(desc_block, count_block) -> (all_desc, all_counts) or None
        rt.set_device(device)
        self.ex = Extractor(self.params, width, height, device)
        self.mt = Matcher(BOW_RATIO, True, device)
        self.caps = self.ex.caps
        self.cap = max(self.caps)
        self.scale = self.ex.GetScaleFactors(0)
        total_refs = self.cap * self.n_cams * world_size
        # HBM scratch for the cross-camera matcher: contiguous reference block + results
        self.d_refs = rt.DeviceBuffer(total_refs * 32)
        self.d_res = [rt.DeviceBuffer(self.cap * 4) for _ in range(3)]
        self.d_scratch = rt.DeviceBuffer(max(Matcher.top2_scratch_bytes(self.cap, total_refs), 16))
        self.prev = None
        self.stream = self.ex.stream

    def close(self):
        self.ex.close(); self.mt.close()

    # images: list of HxW uint8 arrays, or list of (device_ptr, stride) for HBM-resident frames
    def step(self, images, resident=False):
        ex, mt = self.ex, self.mt
        for c, im in enumerate(images):
            if resident:
                ex.upload_device(c, im[0], self.width, self.height, im[1])
            else:
                ex.upload(c, im)
        ex.run()
        per_cam = [ex.download(c) for c in range(self.n_cams)]
        counts = [len(k) for k, _ in per_cam]

        # ---- temporal projection search (a10) on the merged frame
        uright = np.concatenate([synth_uright(k) for k, _ in per_cam]) if sum(counts) else np.zeros(0, np.float32)
        fd = FrameData.from_cameras(per_cam, self.width, self.height, uright)
        n_temporal = 0; match_of = np.zeros(0, np.int32)
        if self.prev is not None and fd.n_total > 0:
            frame = mt.frame(fd)
            q = make_queries(self.prev, self.scale)
            n_temporal, match_of = mt.SearchByProjection(frame, q)
            frame.close()
        self.prev = per_cam

        # ---- cross-camera exhaustive top-2 (a12): each owned camera against every OTHER camera of the rig
        blocks, blk_counts, owner = self._all_descriptors(per_cam)
        n_cross = 0; cross = []
        for c in range(self.n_cams):
            gcam = self.rank * self.n_cams + c
            nq = counts[c]
            off = 0
            for b, (dptr, n) in enumerate(zip(blocks, blk_counts)):
                if owner[b] == gcam or n == 0:
                    continue
                rt._L().orb_memcpy_d2d(self.d_refs.ptr + off * 32, dptr, n * 32, self.stream)
                off += n
            if nq == 0:
                cross.append((np.zeros(0, np.int32),) * 3); continue
            Matcher.hamming_top2_device(ex.device_descriptors(c), nq, self.d_refs.ptr, off, self.d_res[0].ptr,
                                        self.d_res[1].ptr, self.d_res[2].ptr, self.d_scratch.ptr, self.stream)
            bi = self.d_res[0].download(np.int32, nq, self.stream)
            bd = self.d_res[1].download(np.int32, nq, self.stream)
            sd = self.d_res[2].download(np.int32, nq, self.stream)
            cross.append((bi, bd, sd))
            n_cross += int(accept_cross(bd, sd).sum())
        return dict(per_cam=per_cam, counts=counts, n_temporal=n_temporal, match_of_feature=match_of, cross=cross,
                    n_cross=n_cross)

    def _all_descriptors(self, per_cam):
        """[(device pointer, count)] of every camera's descriptor block in global camera order."""
        if self.world == 1 or self.gather is None:
            return ([self.ex.device_descriptors(c) for c in range(self.n_cams)], [len(k) for k, _ in per_cam],
                    list(range(self.n_cams)))
        return self.gather(self, per_cam)
