"""Camera frames cut from real photographs (round 6).  TEST INFRASTRUCTURE.

The reference only ever sees photographs (Examples/RGB-D/rgbd_tum.cc:108-132 -> Tracking::GrabImageRGBD, src/Tracking.cc:236-287:
cvtColor to grey, then Frame::Frame).  None ship with it; three natural photographs are part of this image's Python packages
(scikit-learn's china.jpg and flower.jpg, matplotlib's grace_hopper.jpg).  tests/golden/make_natural.py decodes them ONCE in the build
container, converts to grey with cvtColor's 8-bit formula and stores the grey planes in tests/golden/natural_photos.npz; everything
below is integer arithmetic on those stored planes, so every box derives the same bytes.

  canvas(photo, k)   the grey plane enlarged k times (integer bilinear, pixel centres aligned, round to nearest) and mirrored once
                     along both axes -> a (2kH) x (2kW) plane without seams in value
  frame(...)         a W x H window of that canvas: timestep t slides the window by (-3, -1) px (content moves by MOTION = (+3, +1) as
                     in synth.image), camera c of a rig is the same window moved right by c * baseline px (rigidly mounted cameras with
                     overlapping views, OtherFiles/calibration.txt), plus per-pixel sensor noise in [-noise, noise] that differs
                     per camera and timestep
"""
import hashlib
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PHOTOS_NPZ = os.path.join(HERE, "golden", "natural_photos.npz")
PHOTOS = ("china", "flower", "hopper")
SIZES = ((640, 480, 1000), (1280, 720, 2000), (1920, 1080, 4000))   # (width, height, features per camera) = configs[1], [2], [4]
T_MAX = 16          # timesteps a sequence may have
BASELINE = 24       # px between neighbouring cameras of a rig (level 0)

_M32 = np.uint64(0xFFFFFFFF)


def rgb_to_grey(rgb):
    """cv::cvtColor(RGB2GRAY) on 8-bit data: (R*4899 + G*9617 + B*1868 + 8192) >> 14 (OpenCV 2.4/3.2 color.cpp, RGB2Gray<uchar>)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _hash32(x):
    x = np.asarray(x, np.uint64) & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


_photos = None


def photos():
    """{name: uint8 HxW grey plane} from the committed fixture."""
    global _photos
    if _photos is None:
        z = np.load(PHOTOS_NPZ)
        _photos = {k: z[k] for k in PHOTOS}
    return _photos


def _upsample_axis(a, k, axis):
    """k-fold linear enlargement along one axis, pixel centres aligned: output j sits at (2j + 1 - k) / 2k of the input grid;
    the value is kept times 2k (exact integers), edge samples replicate."""
    n = a.shape[axis]
    j = np.arange(n * k, dtype=np.int64)
    pos = 2 * j + 1 - k                       # in units of 1 / 2k
    i0 = np.floor_divide(pos, 2 * k)
    f = pos - i0 * 2 * k                      # 0 .. 2k-1
    i1 = np.clip(i0 + 1, 0, n - 1); i0 = np.clip(i0, 0, n - 1)
    a0 = np.take(a, i0, axis=axis); a1 = np.take(a, i1, axis=axis)
    shape = [1, 1]; shape[axis] = -1
    f = f.reshape(shape)
    return a0 * (2 * k - f) + a1 * f


_canvas = {}


def canvas(photo, k):
    key = (photo, k)
    if key not in _canvas:
        g = photos()[photo].astype(np.int64)
        if k > 1:
            g = _upsample_axis(_upsample_axis(g, k, 1), k, 0)        # values times 4k^2
            g = (g + 2 * k * k) // (4 * k * k)
        g = g.astype(np.uint8)
        g = np.concatenate([g, g[:, ::-1]], axis=1)
        g = np.concatenate([g, g[::-1, :]], axis=0)
        _canvas[key] = g
    return _canvas[key]


def enlargement(photo, width, height, n_cams=2, baseline=BASELINE):
    """smallest k whose mirrored canvas holds every window of a T_MAX-step sequence of the rig"""
    ph, pw = photos()[photo].shape
    k = 1
    while 2 * k * pw < width + 3 * T_MAX + (n_cams - 1) * baseline + 8 or 2 * k * ph < height + T_MAX + 8:
        k += 1
    return k


def frame(photo, cam, t, width, height, noise=2, baseline=BASELINE, n_cams=2):
    """uint8 HxW frame t (< T_MAX) of camera `cam` of an n_cams rig looking at `photo`."""
    assert 0 <= t < T_MAX
    k = enlargement(photo, width, height, n_cams, baseline)
    cv = canvas(photo, k)
    x0 = 4 + 3 * (T_MAX - 1 - t) + cam * baseline
    y0 = 4 + (T_MAX - 1 - t)
    img = cv[y0:y0 + height, x0:x0 + width]
    assert img.shape == (height, width)
    if not noise:
        return np.ascontiguousarray(img)
    seed = (PHOTOS.index(photo) * 7919 + cam * 100003 + t * 1013 + width) * 2654435761
    idx = np.arange(width * height, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFF)
    nz = (_hash32(idx) % np.uint32(2 * noise + 1)).astype(np.int32).reshape(height, width) - noise
    return np.clip(img.astype(np.int32) + nz, 0, 255).astype(np.uint8)


def rig(photo, t, width, height, n_cams=2, noise=2, baseline=BASELINE):
    return [frame(photo, c, t, width, height, noise, baseline, n_cams) for c in range(n_cams)]


# ---- digests of results (the committed vectors of the larger sizes are SHA-256 of the bytes a test would compare)
def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def step_digests(r):
    """one digest per part of a step result (tests/oracle_pipeline.py: assert_same_step compares exactly these)"""
    return dict(kps=sha(r["kps"]), desc=sha(r["desc"]), stereo=sha(r["uright"], r["depth"]),
                temporal=sha(np.asarray(r["match_of_feature"], np.int32)),
                cross=sha(*[np.asarray(a, np.int32) for a in r["cross"]]),
                counts=np.array(list(r["counts"]) + [r["n_temporal"], r["n_cross"]], np.int32))


SEQ_STEPS = 4
