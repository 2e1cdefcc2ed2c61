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
import ctypes as C
import numpy as np
from . import _lib
from ._lib import QUERY_DTYPE
from .extractor import Extractor
from .matcher import Matcher, TH_LOW
from . import rt
from .frontend import SKIP_CROSS, NO_QUERY_RECORDS

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
    """Last-frame map points projected into the current frame (what src/ORBmatcher.cc:3502-3552 computes on the host)
    under the synthetic stream's known motion.  prev = (keypoints_total, descriptors_total, depth_total, cam_of).
    Host-only C helper (orbm_queries_from_motion), shared by the GPU leg and the CPU-oracle leg of the benchmark."""
    from . import _lib
    from ._lib import ptr
    k, d, depth, cam_of = prev[:4]
    un = prev[4:6] if len(prev) >= 6 else (None, None)   # undistorted positions (mvKeysUn), when a calibration is in use
    n = len(k)
    q = np.empty(n, QUERY_DTYPE)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    unx = np.ascontiguousarray(un[0], np.float32) if un[0] is not None else None
    uny = np.ascontiguousarray(un[1], np.float32) if un[1] is not None else None
    _lib.check(_lib.lib().orbm_queries_from_motion(ptr(k), ptr(d), ptr(depth), ptr(cam_of), n, MOTION[0], MOTION[1], TH_PROJ,
                                                   ptr(sf), MBF, ptr(q), ptr(unx) if unx is not None else None,
                                                   ptr(uny) if uny is not None else None))
    return q


def accept_cross(best_dist, second_dist):
    """SearchByBoW acceptance (src/ORBmatcher.cc:324-327): best <= TH_LOW and best < ratio * second (float compare)."""
    return (best_dist <= TH_LOW) & (best_dist.astype(np.float32) < np.float32(BOW_RATIO) * second_dist.astype(np.float32))


class FrontEnd:
    """The front end of the cameras owned by this process (one process per GPU), on the native orbf_step entry."""

    def __init__(self, params_per_cam, width, height, device=0, rank=0, world_size=1, gather=None, global_cams=None, calib=None):
        from .frontend import NativeFrontEnd
        from .extractor import tables
        self.params = list(params_per_cam); self.n_cams = len(self.params)
        self.width, self.height = width, height
        self.rank, self.world = rank, world_size
        self.gather = gather                      # DescriptorExchange (multi-GPU) or None
        self.global_cams = global_cams or list(range(rank * self.n_cams, (rank + 1) * self.n_cams))
        rt.set_device(device)
        # (where a multi-GPU exchange runs is the native handle's decision, taken when the exchange is set up -- orbf_exchange_placement)
        self.fe = NativeFrontEnd(self.params, width, height, device, ahead_depth=0)
        self.fe.configure(MBF, 100, True)
        if calib is not None:
            self.fe.set_calibration(calib)   # (fx, fy, cx, cy, k1, k2, p1, p2[, k3]): undistortion as the reference's Frame does it
        self.caps = [p.nfeatures + 4 * p.nlevels for p in self.params]
        self.cap = max(self.caps)
        self.scale = tables(self.params[0])["scale"]
        self.depth_host = [synth_depth_image(g, width, height) for g in self.global_cams]
        self.depth_dev = []
        for c, d in enumerate(self.depth_host):
            b = rt.DeviceBuffer(d.nbytes); b.upload(d); self.depth_dev.append(b)
            self.fe.set_depth(c, b.ptr, width)
        # (the uploads above are synchronous; no device-wide synchronisation here: another thread's front end may be
        # capturing its launch chain at this moment, and hipDeviceSynchronize is refused while any capture is open)
        self.copy_results = True   # False: results are views of the native pinned buffers (valid until the next step)
        # thin views of the composed handles (stage timings, output binding, block-list cross matching)
        self.ex = _Handle(Extractor, self.fe.extractor_handle, self.params)
        self.mt = _Handle(Matcher, self.fe.matcher_handle)
        self.stream = self.ex.stream

    def close(self):
        self.fe.close()

    def enable_native_exchange(self, dist, device):
        """Collective: switch the multi-GPU exchange to the native form (RCCL's C API from inside the step, no Python or torch
        call per step).  torch.distributed only carries the communicator id once.  Returns False -- on every rank -- if any
        rank could not set it up (the torch.distributed exchange stays in charge then)."""
        import torch
        world, rank = dist.get_world_size(), dist.get_rank()
        uid = torch.zeros(128, dtype=torch.uint8, device=device)
        ok = 1
        try:
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(self.fe.exchange_unique_id()), dtype=torch.uint8))
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.broadcast(flag, 0)
        if int(flag.item()) == 0:
            return False
        dist.broadcast(uid, 0)
        try:
            self.fe.exchange_init(bytes(uid.cpu().numpy().tobytes()), world, rank)
            ok = 1
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            self.fe.exchange_shutdown()
            return False
        self.native_exchange = True
        return True

    def enable_peer_exchange(self, dist, group=None):
        """Collective: the multi-GPU exchange as direct writes between the rank PROCESSES (orbf_exchange_peer_*: IPC-mapped arenas, no
        RCCL) -- one process per GPU, or several per GPU where there are fewer GPUs than ranks.  torch.distributed (any backend: gloo
        will do) only carries the 64-byte handles once.  Returns False -- on every rank -- if any rank could not set it up."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        try:
            mine = self.fe.exchange_peer_export(world, rank)
        except Exception as e:   # noqa: BLE001
            mine = repr(e)
        handles = [None] * world
        dist.all_gather_object(handles, mine, group=group)
        ok = all(isinstance(h, bytes) for h in handles)
        if ok:
            try:
                self.fe.exchange_peer_open(handles)
            except Exception:    # noqa: BLE001
                ok = False
        oks = [None] * world
        dist.all_gather_object(oks, ok, group=group)
        if not all(oks):
            self.fe.exchange_shutdown()
            return False
        self.native_exchange = True
        return True

    def reset(self):
        self.fe.reset()

    # images: list of HxW uint8 arrays, or list of (device_ptr, stride) for HBM-resident frames
    def announce(self, images, resident=False):
        """Announce the images of a future step (orbf_prefetch; a FIFO, at most three steps ahead): their extraction runs next to
        the matching of the steps before; those steps must then pass exactly these images, in order."""
        if type(images) is self.fe._arr_type:
            pass           # (prepared by prepare())
        elif resident:     # (ptr, stride[, generation]): HBM-resident (True) or page-locked host memory ("pinned": copied H2D inside the step)
            images = [(im[0], self.width, self.height, im[1], 0 if resident == "pinned" else 1, im[2] if len(im) > 2 else 0) for im in images]
        self.fe.prefetch(images)

    def prepare(self, images, resident=True):
        """Marshal the (ptr, stride[, generation]) tuples of HBM-resident (True) or page-locked ("pinned") frames once; step() and
        its next_images take the result in place of the tuples (a ring of buffers is prepared slot by slot)."""
        od = 0 if resident == "pinned" else 1
        return self.fe.prepare([(im[0], self.width, self.height, im[1], od, im[2] if len(im) > 2 else 0) for im in images])

    def step(self, images, resident=False, next_images=None):
        """next_images: shorthand for announce(next_images) before the step."""
        native = getattr(self, "native_exchange", False)
        distributed = self.world > 1 and self.gather is not None and not native
        if not distributed:
            # one native call per timestep: announce + step + the count of accepted cross-camera matches (orbf_step_motion_ahead) --
            # also with the native multi-GPU exchange (the step then returns the rig-wide top-2 of this rank's features)
            AT = self.fe._arr_type
            od = 0 if resident == "pinned" else 1
            # (prepared by prepare(): nothing to marshal; the two arguments are converted independently -- a prepared `images`
            # next to `next_images` still given as (ptr, stride[, generation]) tuples is a legal mix)
            if resident and type(images) is not AT:
                images = [(im[0], self.width, self.height, im[1], od, im[2] if len(im) > 2 else 0) for im in images]
            if resident and next_images is not None and type(next_images) is not AT:
                next_images = [(im[0], self.width, self.height, im[1], od, im[2] if len(im) > 2 else 0) for im in next_images]
            # (the binding never hands out the query records: they stay on the device)
            r = self.fe.step_ahead(images, next_images, (MOTION[0], MOTION[1], TH_PROJ), TH_LOW, BOW_RATIO, flags=NO_QUERY_RECORDS,
                                   copy=self.copy_results)
            if r["n_cross"] is None and "cross" in r:
                bi, bd, sd = r["cross"]
                bd = np.ascontiguousarray(bd, np.int32); sd = np.ascontiguousarray(sd, np.int32)
                r["n_cross"] = _lib.lib().orbm_count_ratio_accepted(_lib.ptr(bd), _lib.ptr(sd), len(bd), TH_LOW, BOW_RATIO)
            if native and self.copy_results:   # (not in the timed loop: the gathered trailers name this rank's counts)
                nc = self.n_cams
                assert r["rig_counts"][self.rank * nc:(self.rank + 1) * nc] == r["counts"]
            return r
        if next_images is not None:
            self.announce(next_images, resident)
        if resident and type(images) is not self.fe._arr_type:
            images = [(im[0], self.width, self.height, im[1], 0 if resident == "pinned" else 1, im[2] if len(im) > 2 else 0) for im in images]
        # queries = the previous step's features under the stream's known motion, built natively (orbf_step_motion;
        # same arithmetic as make_queries, which the oracle leg uses)
        if True:
            # the exchange through torch.distributed (MORB_NATIVE_EXCHANGE=0): one all-gather per timestep.  When the step's descriptor block is final already at begin (its extraction ran
            # ahead), the collective and the cross-camera matching are enqueued next to the step's own matching; otherwise
            # they follow the step.  Every rank issues exactly one collective per step either way.
            # (when the block is final before the step is even begun, the collective is started first of all)
            ahead = self.fe.peek_block(images)
            if ahead is not None:
                self.gather.gather_ahead(self, ahead)
            early = self.fe.begin(images, None, SKIP_CROSS, motion=(MOTION[0], MOTION[1], TH_PROJ))
            assert early or ahead is None
            self.early_exchanges = getattr(self, "early_exchanges", 0) + int(early)
            if early:
                self.gather.enqueue(self)
            r = self.fe.end(copy=self.copy_results)
            bi, bd, sd, cnts = self.gather.collect(self, views=not self.copy_results) if early else self.gather(self)
            assert cnts[self.rank * self.n_cams:(self.rank + 1) * self.n_cams] == r["counts"]
            r["cross"] = (bi, bd, sd)
        bi, bd, sd = r["cross"]
        bd = np.ascontiguousarray(bd, np.int32); sd = np.ascontiguousarray(sd, np.int32)
        r["n_cross"] = _lib.lib().orbm_count_ratio_accepted(_lib.ptr(bd), _lib.ptr(sd), len(bd), TH_LOW, BOW_RATIO)
        return r


class _Handle:
    """Borrowed view of a native handle owned by the front end: exposes the wrapper class' methods without owning it."""

    def __init__(self, cls, handle, params=None):
        self._cls = cls
        self._h = C.c_void_p(handle)
        if params is not None:
            self.params = list(params); self.n_cams = len(self.params)

    def __getattr__(self, name):
        attr = getattr(self._cls, name)
        if isinstance(attr, property):
            return attr.fget(self)
        if callable(attr):
            return lambda *a, **k: attr(self, *a, **k)
        return attr
