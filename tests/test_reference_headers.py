"""The drop-in must compile against the reference's OWN headers: `-fsyntax-only -DMORB_USE_REFERENCE_TYPES` over
the three host sources with /root/reference/include used IN PLACE (nothing of the reference is copied; OpenCV, which this image
lacks, is stood in for by host/cv_compat.h through the include shim host/cv_shim/).  Two arrangements:

  A. the reference tree untouched: its Frame.h / KeyFrame.h / MapPoint.h pull in its own ORBextractor.h and ORBVocabulary.h;
     our ORBmatcher.cc / ORBextractor.cc / ORBVocabulary.cc must compile beside them;
  B. the integration INTEGRATION.md describes: our ORBextractor.h, ORBmatcher.h and ORBVocabulary.h take the place of the
     reference's three (a directory of symbolic links: the reference's headers where they lie, ours for those three), and besides
     our sources the reference's own consumers of the classes that need nothing but OpenCV core -- src/KeyFrameDatabase.cc
     (ORBVocabulary::score / size), src/MapPoint.cc (ORBmatcher::DescriptorDistance), src/Map.cc -- must compile UNCHANGED.
     (Frame.cc / KeyFrame.cc / Tracking.cc need Eigen and Pangolin, which are absent: not attempted, not faked.)

Skipped when /root/reference is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multi_orb_slam_amd", "host")
REF = "/root/reference"
OURS = ["ORBmatcher.cc", "ORBextractor.cc", "ORBVocabulary.cc"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="reference checkout not present")


def _syntax_only(src, includes):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-ffp-contract=off", "-DMORB_USE_REFERENCE_TYPES"]
    for inc in includes:
        cmd += ["-I", inc]
    cmd.append(src)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    errors = [ln for ln in p.stderr.splitlines() if "error" in ln]
    return p.returncode, errors


@pytest.mark.parametrize("src", OURS)
def test_host_sources_compile_against_the_reference_headers_in_place(src):
    rc, errors = _syntax_only(os.path.join(HOST, src),
                              [HOST, os.path.join(HOST, "cv_shim"), os.path.join(REF, "include"), REF, os.path.join(ROOT, "include")])
    assert rc == 0 and not errors, "\n".join(errors[:20])


def _replaced_tree(tmp_path):
    inc = tmp_path / "include"
    inc.mkdir()
    for name in os.listdir(os.path.join(REF, "include")):
        os.symlink(os.path.join(REF, "include", name), inc / name)
    for name in ("ORBextractor.h", "ORBmatcher.h", "ORBVocabulary.h"):       # ours take the place of the reference's
        os.unlink(inc / name)
        os.symlink(os.path.join(HOST, name), inc / name)
    for name in ("cv_compat.h", "slam_types.h"):
        os.symlink(os.path.join(HOST, name), inc / name)
    return str(inc)


@pytest.mark.parametrize("src", OURS + ["ref:src/KeyFrameDatabase.cc", "ref:src/MapPoint.cc", "ref:src/Map.cc"])
def test_replaced_headers_serve_our_sources_and_the_reference_consumers(tmp_path, src):
    inc = _replaced_tree(tmp_path)
    path = os.path.join(REF, src[4:]) if src.startswith("ref:") else os.path.join(HOST, src)
    rc, errors = _syntax_only(path, [inc, os.path.join(HOST, "cv_shim"), REF, os.path.join(ROOT, "include")])
    assert rc == 0 and not errors, "\n".join(errors[:20])


def test_stand_in_types_are_the_reference_types():
    """slam_types.h must declare the two index maps with the reference's container type (include/Frame.h:256,261,
    include/KeyFrame.h:243,248)."""
    import re
    want = re.compile(r"std::unordered_map<size_t,\s*int>\s+keypoint_to_cam")
    for hdr in ("Frame.h", "KeyFrame.h"):
        assert want.search(open(os.path.join(REF, "include", hdr)).read()), hdr
    assert len(want.findall(open(os.path.join(HOST, "slam_types.h")).read())) == 2
