"""Frame::UndistortKeyPoints / ComputeImageBounds (reference src/Frame.cc:673-778): cv::undistortPoints(pts, K, dist,
noArray(), K) of OpenCV 2.4.x / 3.2 restated in the oracle.  Pins: an independent pure-Python restatement (Python floats
are IEEE doubles, evaluated left to right without contraction), the copy branch for k1 == 0, the inverse property against
the forward radial-tangential model, and the reference's own calibration (OtherFiles/multi.yaml:7-16)."""
import struct
import numpy as np
import oracle

MULTI_YAML = (522.6309776476671, 521.2605910670696, 325.2888863142117, 234.2819617198055,      # fx fy cx cy
              -0.01682098888100379, -0.06786658266284684, -0.008326822638637795, 0.003629888382443059, 0.0)   # k1 k2 p1 p2 k3


def f32(v):
    return struct.unpack("<f", struct.pack("<f", v))[0]


def py_undistort(calib, xs, ys):
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [f32(c) for c in calib]      # CV_32F parameters promoted to double
    ifx, ify = 1.0 / fx, 1.0 / fy
    x = (float(xs) - cx) * ifx; y = (float(ys) - cy) * ify
    x0, y0 = x, y
    for _ in range(5):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist; y = (y0 - dy) * icdist
    xx = fx * x + 0.0 * y + cx; yy = 0.0 * x + fy * y + cy; ww = 1.0 / (0.0 * x + 0.0 * y + 1.0)
    return f32(xx * ww), f32(yy * ww)


def test_undistort_equals_python_restatement_bit_for_bit():
    rng = np.random.RandomState(5)
    x = (rng.rand(4000) * 640).astype(np.float32); y = (rng.rand(4000) * 480).astype(np.float32)
    x[:4] = [0, 640, 0, 640]; y[:4] = [0, 0, 480, 480]
    ux, uy = oracle.undistort_points(MULTI_YAML, x, y)
    for i in range(len(x)):
        ex, ey = py_undistort(MULTI_YAML, x[i], y[i])
        assert ux[i] == np.float32(ex) and uy[i] == np.float32(ey), i
    assert np.abs(ux - x).max() > 0.5           # the reference calibration moves border points by more than half a pixel


def test_k1_zero_is_a_plain_copy():
    x = np.array([0.0, 1.5, 639.0], np.float32); y = np.array([0.0, 2.5, 479.0], np.float32)
    for calib in (None, (500, 500, 320, 240, 0.0, -0.05, 0.001, 0.001, 0.0)):   # only k1 decides (src/Frame.cc:676)
        ux, uy = oracle.undistort_points(calib, x, y)
        assert ux.tobytes() == x.tobytes() and uy.tobytes() == y.tobytes()
        assert oracle.image_bounds(calib, 640, 480) == (0.0, 0.0, 640.0, 480.0)


def test_undistort_inverts_the_forward_model():
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = MULTI_YAML
    rng = np.random.RandomState(9)
    xu = rng.rand(2000) * 640; yu = rng.rand(2000) * 480              # ideal (undistorted) pixel positions
    xn = (xu - cx) / fx; yn = (yu - cy) / fy
    r2 = xn * xn + yn * yn
    rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    ux, uy = oracle.undistort_points(MULTI_YAML, (xd * fx + cx).astype(np.float32), (yd * fy + cy).astype(np.float32))
    assert np.abs(ux - xu).max() < 2e-3 and np.abs(uy - yu).max() < 2e-3


def test_image_bounds_of_the_reference_calibration():
    minx, miny, maxx, maxy = oracle.image_bounds(MULTI_YAML, 640, 480)
    cx = [py_undistort(MULTI_YAML, a, b) for a, b in ((0, 0), (640, 0), (0, 480), (640, 480))]
    assert minx == min(cx[0][0], cx[2][0]) and maxx == max(cx[1][0], cx[3][0])
    assert miny == min(cx[0][1], cx[1][1]) and maxy == max(cx[2][1], cx[3][1])
    # barrel distortion (k1, k2 < 0): the undistorted image is larger than the sensor on every side, by tens of pixels
    assert -30 < minx < 0 and -30 < miny < 0 and 640 < maxx < 670 and 480 < maxy < 510
