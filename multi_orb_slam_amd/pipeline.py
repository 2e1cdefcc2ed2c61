"""One front-end timestep over the C ABI: N-camera extract -> device-resident frame assembly (merge, depth -> virtual
right coordinate, 64x48 grid) -> temporal SearchByProjection -> cross-camera exhaustive top-2.  This is what bench.py
times, what smoke() runs once, and what each rank of the multi-GPU driver runs on its own cameras (the cross-camera
step then matches against every rank's descriptors, exchanged with one all-gather).

The step mirrors the reference's per-frame front end: Frame::Frame (src/Frame.cc:148-288: two extractor calls, the
`_total` merge, ComputeStereoFromRGBD, AssignFeaturesToGrid), TrackWithMotionModel's SearchByProjection
(src/Tracking.cc:1267 -> src/ORBmatcher.cc:3448) and the brute-force top-2 inner loop with SearchByBoW's acceptance rule
(src/ORBmatcher.cc:287-327).  The projection of last-frame points is host arithmetic in the reference too; here the
synthetic stream's known motion plays the motion model.  Keypoints, descriptors and depths come back to the host every
step (the rest of a SLAM system needs them there); everything between extraction and the match lists stays in HBM.
"""
import numpy as np
from ._lib import QUERY_DTYPE
from .extractor import Extractor
from .matcher import Matcher, TH_LOW
from . import rt

MOTION = (3.0, 1.0)      # synthetic stream: frame t+1 = frame t translated by (3, 1) px
TH_PROJ = 15.0           # TrackWithMotionModel search radius parameter (src/Tracking.cc:1267)
BOW_RATIO = 0.7          # ORBmatcher(0.7, true).SearchByBoW (src/Tracking.cc:1076)
MBF = 40.0               # Camera.bf (OtherFiles/multi.yaml)


def synth_depth_image(cam, width, height):
    """Static synthetic depth map in metres (float32), with holes (0 = no depth) like a real RGB-D sensor."""
    y, x = np.mgrid[0:height, 0:width]
    d = (1.0 + ((x * 31 + y * 17 + cam * 7) % 64) / 8.0).astype(np.float32)
    d[((x // 8 + y // 8 + cam) % 11) == 0] = 0.0
    return d


def make_queries(prev, scale_factors):
    """Last-frame map points projected into the current frame (what src/ORBmatcher.cc:3502-3552 computes on the host).
    prev = (keypoints_total, descriptors_total, depth_total, cam_of)."""
    k, d, depth, cam_of = prev
    q = np.zeros(len(k), QUERY_DTYPE)
    u = k["x"] + np.float32(MOTION[0])
    q["u"] = u; q["v"] = k["y"] + np.float32(MOTION[1])
    q["radius"] = np.float32(TH_PROJ) * scale_factors[k["octave"]]
    inv = np.where(depth > 0, np.float32(1.0) / np.maximum(depth, np.float32(1e-6)), np.float32(0)).astype(np.float32)
    q["ur"] = u - np.float32(MBF) * inv
    q["min_level"] = k["octave"] - 1; q["max_level"] = k["octave"] + 1
    q["cam"] = cam_of; q["blocks"] = 1; q["angle"] = k["angle"]; q["desc"] = d
    return q


def accept_cross(best_dist, second_dist):
    """SearchByBoW acceptance (src/ORBmatcher.cc:324-327): best <= TH_LOW and best < ratio * second (float compare)."""
    return (best_dist <= TH_LOW) & (best_dist.astype(np.float32) < np.float32(BOW_RATIO) * second_dist.astype(np.float32))


class FrontEnd:
    """Extractor + matcher for the cameras owned by this process (one process per GPU)."""

    def __init__(self, params_per_cam, width, height, device=0, rank=0, world_size=1, gather=None, global_cams=None):
        self.params = list(params_per_cam); self.n_cams = len(self.params)
        self.width, self.height = width, height
        self.rank, self.world = rank, world_size
        self.gather = gather                      # DescriptorExchange (multi-GPU) or None
        self.global_cams = global_cams or list(range(rank * self.n_cams, (rank + 1) * self.n_cams))
        rt.set_device(device)
        self.ex = Extractor(self.params, width, height, device)
        self.mt = Matcher(BOW_RATIO, True, device)
        self.mt.set_stream(self.ex.stream)        # one stream: frame build and matching are ordered after extraction
        self.caps = self.ex.caps
        self.cap = max(self.caps)
        self.scale = self.ex.GetScaleFactors(0)
        self.depth_host = [synth_depth_image(g, width, height) for g in self.global_cams]
        self.depth_dev = []
        for d in self.depth_host:
            b = rt.DeviceBuffer(d.nbytes); b.upload(d); self.depth_dev.append(b)
        rt.device_sync()
        self.prev = None
        self.stream = self.ex.stream

    def close(self):
        self.mt.close(); self.ex.close()

    # images: list of HxW uint8 arrays, or list of (device_ptr, stride) for HBM-resident frames
    def step(self, images, resident=False):
        ex, mt = self.ex, self.mt
        for c, im in enumerate(images):
            if resident:
                ex.upload_device(c, im[0], self.width, self.height, im[1])
            else:
                ex.upload(c, im)
        ex.run()
        counts = [ex.count(c) for c in range(self.n_cams)]
        cams = [(ex.device_keypoints(c), ex.device_descriptors(c), counts[c], self.depth_dev[c].ptr, self.width)
                for c in range(self.n_cams)]
        frame = mt.frame_from_device(cams, MBF, (0.0, 0.0, float(self.width), float(self.height)))
        kps, desc, uright, depth = frame.download()
        cam_of = np.repeat(np.arange(self.n_cams, dtype=np.int32), counts)

        n_temporal = 0; match_of = np.zeros(0, np.int32)
        if self.prev is not None and frame.data.n_total > 0:
            q = make_queries(self.prev, self.scale)
            n_temporal, match_of = mt.SearchByProjection(frame, q)
        self.prev = (kps, desc, depth, cam_of)

        if self.world == 1 or self.gather is None:
            bi, bd, sd = mt.cross_top2(frame)
        else:
            ptrs, cnts = self.gather(self, counts)
            bi, bd, sd = mt.cross_top2_blocks(ptrs, cnts, self.rank * self.n_cams, self.n_cams)
            assert cnts[self.rank * self.n_cams:(self.rank + 1) * self.n_cams] == counts
        n_cross = int(accept_cross(bd, sd).sum())
        frame.close()
        return dict(kps=kps, desc=desc, uright=uright, depth=depth, counts=counts, n_temporal=n_temporal,
                    match_of_feature=match_of, cross=(bi, bd, sd), n_cross=n_cross)
