"""Sanitizer builds of the code that runs on the host (GPU sanitizers do not exist on this pool):

* tests/san/san_oracle     ASan + UBSan over the CPU oracle and the product's host quadtree (csrc/octree.cpp) -- whole extractions on many
                           sizes / pitches / kinds of image, the two quadtrees against each other, the matcher restatements;
* tests/san/tsan_rendezvous  TSan over the loopback exchange's rendezvous (csrc/loop_rendezvous.h) incl. members leaving mid-round;
* host/test_host_san       ASan + UBSan over the C++ host classes and cv_compat.h: the cv::gemm known answers, the pose algebra, and the
                           three search drivers on the GPU tests' own cases up to the first device call (no GPU here: the call reports
                           "no device", after the frames, index tables and queries have been built and flattened).

A report of any sanitizer fails the test; so does a build that does not compile."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")
HOST = os.path.join(ROOT, "multi_orb_slam_amd", "host")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98",
           TSAN_OPTIONS="halt_on_error=1:exitcode=97")
MARKS = ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "WARNING: ThreadSanitizer")


def _run(cmd, ok_codes=(0,), timeout=600):
    p = subprocess.run(cmd, capture_output=True, text=True, env=ENV, timeout=timeout)
    out = p.stdout + p.stderr
    assert not any(mk in out for mk in MARKS), out[-4000:]
    assert p.returncode in ok_codes, (p.returncode, out[-2000:])
    return out


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", SAN], timeout=900)
    subprocess.check_call(["make", "-s", "-C", HOST, "san"], timeout=900)


def test_oracle_and_host_quadtree_under_asan_ubsan(built):
    assert "san_oracle: 0 failures" in _run([os.path.join(SAN, "san_oracle")])


def test_loopback_rendezvous_under_tsan(built):
    assert "tsan_rendezvous: 0 failures" in _run([os.path.join(SAN, "tsan_rendezvous")])


def test_host_classes_pose_algebra_under_asan_ubsan(built):
    b = os.path.join(HOST, "test_host_san")
    assert "gemm: 0 known answers wrong" in _run([b, "gemm"])
    assert "0 differing floats" in _run([b, "rt", "50000"])


@pytest.mark.parametrize("which", ["match", "f4", "bow"])
def test_host_search_drivers_up_to_the_device_call_under_asan_ubsan(built, which, tmp_path, monkeypatch):
    """The GPU tests of the C++ classes (tests/test_gpu_host_cpp.py) write their case file and start host/test_host; here the same case
    goes to the sanitized driver.  Without a GPU every search stops at its first device call -- behind the host code that reads the
    reference's containers (unordered_map lookups, index tables), flattens frames and builds the query records."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the plain driver runs these cases to the end in tests/test_gpu_host_cpp.py")
    import test_gpu_host_cpp as g

    class Stop(Exception):
        pass
    seen = []

    def fake(cmd, **kw):
        seen.append(list(cmd))
        raise Stop
    monkeypatch.setattr(g.subprocess, "check_call", fake)
    with pytest.raises(Stop):
        if which == "match":
            g.test_cpp_orbmatcher_search_by_projection_overloads(tmp_path, 1)
        elif which == "f4":
            g.test_cpp_remaining_projection_searches(tmp_path, 1)
        else:
            g.test_cpp_vocabulary_and_bow_searches(tmp_path, 1, 0, (0, 0))
    cmd = seen[0]
    assert cmd[1] == which
    _run([os.path.join(HOST, "test_host_san")] + cmd[1:], ok_codes=tuple(range(0, 64)))   # (the drivers' own exit codes: not a signal, not a sanitizer's)
