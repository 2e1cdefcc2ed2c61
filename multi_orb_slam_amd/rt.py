"""ctypes mirror of include/orb_rt.h: device buffers, streams and HIP-event timing without torch."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import check


def _L():
    L = _lib.lib()
    if not getattr(L, "_rt_ready", False):
        vp, sz = C.c_void_p, C.c_size_t
        L.orb_device_name.argtypes = [C.c_int, vp, C.c_int]
        L.orb_malloc.argtypes = [vp, sz]; L.orb_free.argtypes = [vp]
        L.orb_malloc_host.argtypes = [vp, sz]; L.orb_free_host.argtypes = [vp]
        for f in (L.orb_memcpy_h2d, L.orb_memcpy_d2h, L.orb_memcpy_d2d):
            f.argtypes = [vp, vp, sz, vp]
        L.orb_memset.argtypes = [vp, C.c_int, sz, vp]
        L.orb_stream_sync.argtypes = [vp]
        L.orb_event_create.argtypes = [vp]; L.orb_event_destroy.argtypes = [vp]
        L.orb_event_record.argtypes = [vp, vp]; L.orb_event_elapsed_ms.argtypes = [vp, vp, vp]
        L._rt_ready = True
    return L


def device_count():
    return _L().orb_device_count()


def device_name(device=0):
    buf = C.create_string_buffer(128)
    check(_L().orb_device_name(device, buf, 128))
    return buf.value.decode()


def set_device(device):
    check(_L().orb_set_device(device))


class DeviceBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._p = C.c_void_p()
        check(_L().orb_malloc(C.byref(self._p), max(self.nbytes, 16)))

    @property
    def ptr(self):
        return self._p.value if self._p else None

    def upload(self, arr, stream=None, offset=0):
        arr = np.ascontiguousarray(arr)
        check(_L().orb_memcpy_h2d(C.c_void_p(self.ptr + offset), arr.ctypes.data_as(C.c_void_p), arr.nbytes, stream))

    def download(self, dtype, count, stream=None, offset=0):
        out = np.zeros(count, dtype)
        check(_L().orb_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), out.nbytes, stream))
        if stream:
            check(_L().orb_stream_sync(stream))
        return out

    def free(self):
        if getattr(self, "_p", None):
            try:
                _L().orb_free(self._p)
            except Exception:
                pass
            self._p = None   # (not a fresh c_void_p: at interpreter shutdown the ctypes module may be gone already)

    __del__ = free


class PinnedBuffer:
    """Page-locked host memory (orb_malloc_host) with a numpy view: the source of asynchronous H2D image uploads."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._p = C.c_void_p()
        check(_L().orb_malloc_host(C.byref(self._p), max(self.nbytes, 16)))
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p.value), dtype=np.uint8)

    @property
    def ptr(self):
        return self._p.value if self._p else None

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            try:
                _L().orb_free_host(self._p)
            except Exception:
                pass
            self._p = None

    __del__ = free


def memcpy_d2d(dst_ptr, src_ptr, nbytes, stream=None):
    """device-to-device copy between raw pointers (synchronous without a stream)"""
    check(_L().orb_memcpy_d2d(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes, stream))


def stream_sync(stream=None):
    check(_L().orb_stream_sync(stream))


def device_sync():
    check(_L().orb_device_sync())


class Event:
    def __init__(self):
        self._e = C.c_void_p()
        check(_L().orb_event_create(C.byref(self._e)))

    def record(self, stream=None):
        check(_L().orb_event_record(self._e, stream))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(_L().orb_event_elapsed_ms(self._e, stop._e, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "_e", None):
            try:
                _L().orb_event_destroy(self._e)
            except Exception:
                pass
            self._e = C.c_void_p()
