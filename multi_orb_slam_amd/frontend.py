"""ctypes mirror of include/orbf.h: one front-end timestep as one native call (single stream, single final sync)."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import KP_DTYPE, QUERY_DTYPE, FImage, FMotion, FResult, Params, check, ptr

SKIP_CROSS = 1
NO_QUERY_RECORDS = 2      # orbf_step_motion*: do not materialise orbf_result::queries on the host (orbf.h)


def _view(addr, dtype, count, shape=None):
    if not addr or count == 0:
        return np.zeros(0 if shape is None else (0,) + tuple(shape[1:]), dtype)
    nbytes = count * np.dtype(dtype).itemsize * (int(np.prod(shape[1:])) if shape is not None else 1)
    buf = (C.c_char * nbytes).from_address(addr)
    a = np.frombuffer(buf, dtype=dtype)
    return a.reshape(shape) if shape is not None else a


class _LazyStep(dict):
    """Result of a step consumed in place (copy=False): a few scalars are there, everything else -- the per-feature arrays as numpy
    views of the native pinned buffers, the per-camera counts, the host timings -- is built from the native result record when first
    asked for (building a dozen views costs more than the native call itself).  The record and the buffers belong to the front end and
    are reused by its next step: asking for something new after that step is an error, not a stale answer."""
    _LAZY = ("counts", "rig_counts", "host_us", "cross_dist_ptrs", "kps", "desc", "uright", "depth", "un_x", "un_y", "match_of_feature", "cross")

    def __init__(self, scalars, fe, nq):
        super().__init__(scalars)
        self._fe = fe; self._seq = fe._step_seq; self._nq = nq

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        if self._fe._step_seq != self._seq:
            raise RuntimeError("step result consumed in place: %r was first asked for after the front end's next step" % key)
        v = self._fe._lazy(key, self._nq)
        if v is _ABSENT:
            raise KeyError(key)
        self[key] = v
        return v

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        if key not in self._LAZY:
            return False
        if self._fe._step_seq != self._seq:   # (the record now belongs to a later step: no answer is better than a wrong one)
            raise RuntimeError("step result consumed in place: %r was first asked about after the front end's next step" % key)
        if key == "cross":
            return bool(self._fe._res.cross_best_idx)
        return True

    def get(self, key, default=None):
        return self[key] if key in self else default

    def materialise(self):
        for k in self._LAZY:
            if k in self:
                self[k]
        return self


_ABSENT = object()


class NativeFrontEnd:
    def _cached(self, name, addr, dtype, cap, row=None):
        """numpy view over a native pinned buffer, created once per (address) and sliced per step."""
        if not addr:
            return None
        ent = self._views.get((name, addr))  # (the per-feature results alternate between two native buffers)
        if ent is None:
            shape = (cap,) if row is None else (cap, row)
            ent = _view(addr, dtype, cap, shape if row is not None else None)
            self._views[(name, addr)] = ent
        return ent

    def __init__(self, params, max_width, max_height, device=0, ahead_depth=0):
        """ahead_depth: timesteps prefetch() accepts ahead (1..3; 0 = MORB_AHEAD_DEPTH, default 3).  Streams are hardware queues and the
        part serves four side by side; the multi-GPU exchange runs on the matcher's own stream and needs none."""
        self.params = list(params); self.n_cams = len(self.params)
        arr = (Params * self.n_cams)(*[p.c() for p in self.params])
        self._h = C.c_void_p()
        check(_lib.lib().orbf_create_depth(arr, self.n_cams, max_width, max_height, device, ahead_depth, C.byref(self._h)))
        self._res = FResult()
        self._arr_type = FImage * self.n_cams
        self._ncross = C.c_int(0)
        # (the per-step call of a stream: function object and argument references made once)
        self._fn_step_ahead = _lib.lib().orbf_step_motion_ahead
        self._res_ref = C.byref(self._res); self._ncross_ref = C.byref(self._ncross)
        self._views = {}
        self._step_seq = 0
        self._img_cache = {}
        self._motion_cache = {}
        self._motion_refs = {}     # motion -> (byref, FMotion) for step_ahead
        self.cap_total = sum(p.nfeatures + 4 * p.nlevels for p in self.params)
        self._imgs = (FImage * self.n_cams)()
        self._ready = C.c_int(0)

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().orbf_destroy(self._h)
            except Exception:
                pass
            self._h = None

    __del__ = close

    def set_depth(self, cam, d_ptr, stride_floats):
        check(_lib.lib().orbf_set_depth(self._h, cam, C.c_void_p(d_ptr) if d_ptr else None, stride_floats))

    def set_calibration(self, calib):
        """calib = (fx, fy, cx, cy, k1, k2, p1, p2[, k3]) or None (orbf_set_calibration)."""
        from ._lib import Calibration
        if calib is None:
            check(_lib.lib().orbf_set_calibration(self._h, None))
        else:
            c = Calibration(*(list(calib) + [0.0] * (9 - len(calib))))
            check(_lib.lib().orbf_set_calibration(self._h, C.byref(c)))

    def configure(self, mbf=40.0, th_high=100, check_orientation=True):
        check(_lib.lib().orbf_configure(self._h, mbf, th_high, int(check_orientation)))

    @property
    def extractor_handle(self):
        return _lib.lib().orbf_extractor(self._h)

    @property
    def matcher_handle(self):
        return _lib.lib().orbf_matcher(self._h)

    def reset(self):
        check(_lib.lib().orbf_reset(self._h))

    @staticmethod
    def _fill(dst, images):
        keep = []
        for c, im in enumerate(images):
            if isinstance(im, np.ndarray):
                im = np.ascontiguousarray(im, np.uint8); keep.append(im)
                dst[c] = FImage(im.ctypes.data, im.shape[1], im.shape[0], im.strides[0], 0, 0)
            elif im is None:
                dst[c] = FImage(None, 0, 0, 0, 0, 0)
            else:
                # (ptr, width, height, stride[, on_device[, generation]])
                dst[c] = FImage(im[0], im[1], im[2], im[3], 1 if (len(im) < 5 or im[4]) else 0, im[5] if len(im) > 5 else 0)
        return keep

    def export_block(self):
        """(device pointer, bytes, rows) of the last step's descriptor block + count trailer (orbf_export_block)."""
        p = C.c_void_p(); nb = C.c_size_t(); rows = C.c_int()
        check(_lib.lib().orbf_export_block(self._h, C.byref(p), C.byref(nb), C.byref(rows)))
        return p.value, nb.value, rows.value

    def exchange_unique_id(self):
        buf = (C.c_uint8 * 128)()
        check(_lib.lib().orbf_exchange_unique_id(buf))
        return bytes(buf)

    def exchange_init(self, uid, world, rank):
        """Collective over the ranks (orbf_exchange_init): from now on every step all-gathers natively over RCCL."""
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(_lib.lib().orbf_exchange_init(self._h, buf, world, rank))

    def exchange_init_loopback(self, group, world, rank):
        """Join an in-process exchange group (orbf_exchange_init_loopback): `world` front ends on one device, one thread each."""
        check(_lib.lib().orbf_exchange_init_loopback(self._h, group, world, rank))

    def exchange_peer_export(self, world, rank):
        """Step 1 of the peer transport (orbf_exchange_peer_export): allocate this rank's receive arena; -> its IPC handle (bytes) for
        the other ranks."""
        n = int(_lib.lib().orbf_exchange_peer_handle_bytes())
        buf = (C.c_uint8 * n)()
        check(_lib.lib().orbf_exchange_peer_export(self._h, world, rank, buf))
        return bytes(buf)

    def exchange_peer_open(self, handles):
        """Step 2 (orbf_exchange_peer_open): `handles` = every rank's handle in rank order (bytes objects)."""
        blob = b"".join(handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        check(_lib.lib().orbf_exchange_peer_open(self._h, buf))

    def exchange_shutdown(self):
        check(_lib.lib().orbf_exchange_shutdown(self._h))

    @property
    def ahead_depth(self):
        """timesteps prefetch() accepts ahead of the step being matched (orbf_ahead_depth)"""
        return _lib.lib().orbf_ahead_depth(self._h)

    @property
    def exchange_world(self):
        return _lib.lib().orbf_exchange_active(self._h)

    @property
    def exchange_placement(self):
        """0 no exchange, 1 on the matcher's own stream behind the step's search, 3 at the tail of the step's extraction chain
        (orbf_exchange_placement; MORB_EXCHANGE_PLACEMENT = chain | inline when the exchange is set up)"""
        return _lib.lib().orbf_exchange_placement(self._h)

    def debug_exchange_timing(self, on=True):
        check(_lib.lib().orbf_debug_exchange_timing(self._h, 1 if on else 0))

    def debug_exchange_redos(self):
        """steps whose blocks were shipped a second time (orbf_debug_exchange_redos)"""
        return int(_lib.lib().orbf_debug_exchange_redos(self._h))

    def debug_exchange_us(self):
        """(search finished, exchange finished) of the last step, microseconds of device time from the start of its matching (the second
        one is negative when the exchange -- issued with the step's extraction chain -- was over before the matching began)"""
        out = (C.c_float * 2)()
        check(_lib.lib().orbf_debug_exchange_us(self._h, out))
        return float(out[0]), float(out[1])

    def peek_block(self, images):
        """Before begin(): (device pointer, bytes, rows) of the step's export block when it is final already, else None
        (orbf_peek_block).  `images` as for begin()."""
        arr, self._keep_peek = self._image_array(images, self._imgs)
        p = C.c_void_p(); nb = C.c_size_t(); rows = C.c_int()
        check(_lib.lib().orbf_peek_block(self._h, arr, C.byref(p), C.byref(nb), C.byref(rows)))
        return (p.value, nb.value, rows.value) if p.value else None

    def export_features(self):
        """HBM-resident arrays of the last completed step's frame (orbf_export_features) -> _lib.DeviceFeatures."""
        from ._lib import DeviceFeatures
        d = DeviceFeatures()
        check(_lib.lib().orbf_export_features(self._h, C.byref(d)))
        return d

    def _image_array(self, images, slot):
        """ctypes orbf_image array for `images`; HBM-resident frames (tuples) are built once per distinct set and reused (a
        stream cycles through a ring of buffers), host arrays are filled into the scratch array `slot`."""
        if type(images) is self._arr_type:      # prepared by prepare()
            return images, []
        try:
            key = tuple(images)
            arr = self._img_cache.get(key)
        except TypeError:      # numpy arrays are not hashable: host images
            key = arr = None
        if arr is None:
            if key is not None and all(isinstance(im, tuple) for im in images):
                arr = (FImage * self.n_cams)()
                self._fill(arr, images)
                if len(self._img_cache) < 256:
                    self._img_cache[key] = arr
                return arr, []
            arr = slot
            return arr, self._fill(arr, images)
        return arr, []

    def prefetch(self, next_images):
        """Declare the images of the step after the next one (orbf_prefetch): their extraction overlaps the next step's
        matching.  The arrays / device buffers must stay alive and unchanged until the step that consumes them returns."""
        if not hasattr(self, "_next_imgs"):
            self._next_imgs = (FImage * self.n_cams)()
        if not hasattr(self, "_next_keep"):
            self._next_keep = []
        arr, keep = self._image_array(next_images, self._next_imgs)
        if keep:
            self._next_keep = (self._next_keep + [keep])[-4:]   # host arrays stay alive across the steps in between
        check(_lib.lib().orbf_prefetch(self._h, arr))

    def run_stream(self, ring, t0, steps, ahead, announced_upto, motion, th_low, ratio):
        """orbf_run_stream: `steps` timesteps of the synthetic stream in one native call.  ring = list (one entry per ring
        slot) of per-camera image tuples (ptr, width, height, stride, on_device).  -> (stats dict, new announced_upto)."""
        from ._lib import FStreamStats
        key = id(ring)
        cached = getattr(self, "_ring_cache", None)
        if cached is None or cached[0] != key:
            arr = (FImage * (len(ring) * self.n_cams))()
            for t, images in enumerate(ring):
                sub = (FImage * self.n_cams).from_address(C.addressof(arr) + t * self.n_cams * C.sizeof(FImage))
                self._fill(sub, images)
            self._ring_cache = cached = (key, arr, ring)
        st = FStreamStats(); upto = C.c_int(announced_upto)
        mo = FMotion(*motion)
        self._step_seq += 1        # (the native loop reuses the result record and the pinned buffers: earlier in-place results end here)
        check(_lib.lib().orbf_run_stream(self._h, cached[1], len(ring), t0, steps, ahead, C.byref(upto), C.byref(mo), th_low,
                                         C.c_float(ratio), C.byref(st)))
        return (dict(features=st.features, temporal_matches=st.temporal_matches, cross_accepted=st.cross_accepted,
                     digest=st.digest, seconds=st.seconds), upto.value)

    def step(self, images, queries=None, flags=0, copy=True, motion=None):
        """images: [(ptr_or_array, width, height, stride, on_device)] or uint8 arrays.  `queries`: projected map points, or
        `motion` = (du, dv, th): queries built natively from the previous step's features (synthetic-stream driver).
        Returns a dict of numpy arrays (copies by default: the native buffers are reused by the next step)."""
        self.begin(images, queries, flags, motion)
        return self.end(copy)

    def begin(self, images, queries=None, flags=0, motion=None):
        """First half of step() (orbf_step_begin): everything is enqueued, nothing waited for.  Returns True when the step's
        export block is already final, i.e. a multi-GPU exchange may be enqueued before end()."""
        arr, self._keep = self._image_array(images, self._imgs)
        ready = self._ready
        self._step_seq += 1        # every native step reuses the result record: an in-place result of an earlier step is stale from here on
        if motion is not None:
            mo = self._motion_cache.get(motion)
            if mo is None:
                mo = self._motion_cache[motion] = FMotion(*motion)
            check(_lib.lib().orbf_step_motion_begin(self._h, arr, C.byref(mo), flags, C.byref(ready)))
            self._nq = None
        else:
            nq = 0 if queries is None else len(queries)
            if nq:
                queries = np.ascontiguousarray(queries, QUERY_DTYPE)
            self._keep.append(queries)
            check(_lib.lib().orbf_step_begin(self._h, arr, ptr(queries) if nq else None, nq, flags, C.byref(ready)))
            self._nq = nq
        return bool(ready.value)

    def prepare(self, images):
        """The orbf_image array of HBM-resident / page-locked frames (tuples (ptr, width, height, stride, on_device[, generation])),
        built once: step_ahead() takes it as `images` / `next_images` without marshalling it again (a stream that cycles through a
        ring of buffers prepares every slot once).  The buffers must stay alive, as for step()."""
        arr = self._arr_type()
        self._fill(arr, images)
        return arr

    def step_ahead(self, images, next_images, motion, th_low, ratio, flags=0, copy=True):
        """prefetch(next_images) + step(images, motion=...) + the count of cross-camera matches a (th_low, ratio) acceptance keeps,
        in ONE native call (orbf_step_motion_ahead): what a stream-driving host does per timestep.  -> step() result with
        ["n_cross"] (None when the step has no cross-camera distances)."""
        AT = self._arr_type
        if type(images) is AT:       # prepared by prepare(): HBM-resident frames of a ring, marshalled once
            arr = images
        else:
            arr, self._keep = self._image_array(images, self._imgs)
        nxt = None
        if type(next_images) is AT:
            nxt = next_images
        elif next_images is not None:
            if not hasattr(self, "_next_imgs"):
                self._next_imgs = (FImage * self.n_cams)()
                self._next_keep = []
            nxt, keep = self._image_array(next_images, self._next_imgs)
            if keep:
                self._next_keep = (self._next_keep + [keep])[-4:]
        mo = self._motion_refs.get(motion)
        if mo is None:
            if len(self._motion_refs) >= 64:      # (a stream whose motion changes every step: keep the cache small)
                self._motion_refs.clear()
            m_ = FMotion(*motion)
            mo = self._motion_refs[motion] = (C.byref(m_), m_)
        nc = self._ncross
        self._step_seq += 1
        rc = self._fn_step_ahead(self._h, arr, nxt, mo[0], flags, th_low, ratio, self._res_ref, self._ncross_ref)
        if rc:
            check(rc)
        self._nq = None
        r = self._collect(copy)
        r["n_cross"] = nc.value if nc.value >= 0 else None
        return r

    def _lazy(self, key, nq):
        """one field of the last step's native result record (for _LazyStep)"""
        r = self._res; V = self._cached; n = r.n_total; cap = self.cap_total
        if key == "counts":
            return V("counts", r.counts, np.int32, self.n_cams).tolist()
        if key == "rig_counts":
            return V("rigc", r.rig_counts, np.int32, r.rig_cams).tolist() if r.rig_cams else None
        if key == "host_us":
            return tuple(r.host_us)
        if key == "cross_dist_ptrs":
            return (r.cross_best_dist, r.cross_second_dist) if r.cross_best_idx else None
        if key == "kps":
            return V("kps", r.kps, KP_DTYPE, cap)[:n]
        if key == "desc":
            return V("desc", r.desc, np.uint8, cap, 32)[:n]
        if key == "uright":
            return V("ur", r.uright, np.float32, cap)[:n]
        if key == "depth":
            return V("depth", r.depth, np.float32, cap)[:n]
        if key == "un_x":
            return V("unx", r.un_x, np.float32, cap)[:n]
        if key == "un_y":
            return V("uny", r.un_y, np.float32, cap)[:n]
        if key == "match_of_feature":
            return V("match", r.match_of_feature, np.int32, cap)[:n] if nq else np.zeros(0, np.int32)
        if key == "cross":
            if not r.cross_best_idx:
                return _ABSENT
            return (V("x0", r.cross_best_idx, np.int32, cap)[:n], V("x1", r.cross_best_dist, np.int32, cap)[:n],
                    V("x2", r.cross_second_dist, np.int32, cap)[:n])
        return _ABSENT

    def end(self, copy=True):
        """Second half of step() (orbf_step_end): one synchronisation, then the results."""
        check(_lib.lib().orbf_step_end(self._h, C.byref(self._res)))
        return self._collect(copy)

    def _collect(self, copy):
        r = self._res
        nq = r.n_queries if self._nq is None else self._nq
        n = r.n_total
        cap = self.cap_total
        V = self._cached
        if not copy:
            return _LazyStep({"n_temporal": r.nmatches, "gpu_wait_us": r.gpu_wait_us, "n_queries": nq, "n_total": n}, self, nq)
        cp = lambda a: a.copy()
        out = dict(counts=V("counts", r.counts, np.int32, self.n_cams).tolist(), kps=cp(V("kps", r.kps, KP_DTYPE, cap)[:n]),
                   desc=cp(V("desc", r.desc, np.uint8, cap, 32)[:n]), uright=cp(V("ur", r.uright, np.float32, cap)[:n]),
                   depth=cp(V("depth", r.depth, np.float32, cap)[:n]), un_x=cp(V("unx", r.un_x, np.float32, cap)[:n]),
                   un_y=cp(V("uny", r.un_y, np.float32, cap)[:n]), n_temporal=r.nmatches,
                   match_of_feature=cp(V("match", r.match_of_feature, np.int32, cap)[:n]) if nq else np.zeros(0, np.int32),
                   gpu_wait_us=r.gpu_wait_us, host_us=tuple(r.host_us), n_queries=nq)
        out["rig_counts"] = V("rigc", r.rig_counts, np.int32, r.rig_cams).tolist() if r.rig_cams else None
        if r.cross_best_idx:
            out["cross"] = (cp(V("x0", r.cross_best_idx, np.int32, cap)[:n]), cp(V("x1", r.cross_best_dist, np.int32, cap)[:n]),
                            cp(V("x2", r.cross_second_dist, np.int32, cap)[:n]))
        return out
