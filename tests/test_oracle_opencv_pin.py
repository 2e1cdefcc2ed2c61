"""The OpenCV pin: oracle/pin_opencv.cpp compares every OpenCV operator the reference's front end calls
(cv::resize, cv::copyMakeBorder, cv::FAST, cv::GaussianBlur, cv::fastAtan2, cvRound) with the oracle's restatement, and the
reference's own src/ORBextractor.cc -- compiled from where it lies -- with orc_extract, on the photograph frames of tests/natural.py.

This image has no OpenCV, so here the harness is only kept COMPILING (`make -C oracle pin-syntax`: against declarations of the
entry points, nothing defined, nothing linked) and its oracle-side symbols are checked to exist.  On a box with OpenCV 2.4.x / 3.x
(`pkg-config opencv`) and a checkout of the reference:

    make -C oracle pin REF=/path/to/Multi_ORB_SLAM && MORB_REFERENCE=/path/to/Multi_ORB_SLAM python -m pytest tests/test_oracle_opencv_pin.py

runs test_oracle_is_pinned_against_opencv_and_the_reference_extractor; when it passes, `parity` is pinned (DESIGN.md section 2 names, per
operator, the one oracle function to change when it does not)."""
import ctypes
import os
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
REF = os.environ.get("MORB_REFERENCE", "/root/reference")


def _have_opencv():
    if not shutil.which("pkg-config"):
        return None
    for name in ("opencv", "opencv3", "opencv2"):
        if subprocess.run(["pkg-config", "--exists", name]).returncode == 0:
            return name
    return None


def test_pin_harness_compiles_and_its_oracle_symbols_exist():
    r = subprocess.run(["make", "-s", "-C", ORACLE, "pin-syntax", "REF=" + REF], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "error" not in r.stderr and "warning" not in r.stderr, r.stderr
    import oracle
    lib = oracle.lib()
    for sym in ("orc_level_sizes", "orc_resize_linear_u8", "orc_copy_make_border_reflect101", "orc_fast", "orc_gaussian_blur7",
                "orc_fast_atan2", "orc_cv_round", "orc_extract"):
        assert hasattr(lib, sym), sym
    lib.orc_cv_round.argtypes = [ctypes.c_double]
    assert [lib.orc_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999999, 2.5000001)] == [0, 2, 2, 0, -2, 2, 3]   # halves to even


@pytest.mark.skipif(_have_opencv() is None or not os.path.isfile(os.path.join(REF, "src", "ORBextractor.cc")),
                    reason="needs OpenCV 2.4.x / 3.x (pkg-config opencv) and a checkout of the reference: not in this image")
def test_oracle_is_pinned_against_opencv_and_the_reference_extractor(tmp_path):
    import natural
    r = subprocess.run(["make", "-s", "-C", ORACLE, "pin", "REF=" + REF, "OPENCV_PC=" + _have_opencv()], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    args = []
    for photo in natural.PHOTOS:
        for (w, h, nf) in natural.SIZES:
            for cam in range(2):
                img = natural.frame(photo, cam, 0, w, h)
                p = tmp_path / ("%s_%dx%d_cam%d.raw" % (photo, w, h, cam))
                p.write_bytes(struct.pack("<2i", w, h) + img.tobytes())
                args.append("%d:%s" % (nf, p))
    r = subprocess.run([os.path.join(ORACLE, "_ref", "pin_opencv")] + args, capture_output=True, text=True, timeout=1800)
    print(r.stdout)
    failing = [ln for ln in r.stdout.splitlines() if " FAIL " in ln]
    assert r.returncode == 0 and not failing, "\n".join(failing[:20])
    assert "parity pinned" in r.stdout
