"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls: no GPU here)."""
import ctypes
import glob
import os
import re
import multi_orb_slam_amd as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(orb[xmfv]?_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(m.LIB_PATH)
    fns = declared_functions()
    assert len(fns) >= 80
    for f in fns:
        assert hasattr(lib, f), f


def test_no_device_is_a_loud_error_not_a_fallback():
    # in the CPU container there is no GPU: every handle constructor must fail with ORB_E_NO_DEVICE
    import torch
    if torch.cuda.is_available():
        return
    for ctor in (lambda: m.Matcher(), lambda: m.Extractor(m.ExtractorParams(), 640, 480)):
        try:
            ctor()
        except m.OrbError as e:
            assert e.code == -4 and "no CPU path" in str(e)
        else:
            raise AssertionError("constructor succeeded without a GPU")


def test_keypoint_and_query_layouts():
    assert m.KP_DTYPE.itemsize == 28 and m.QUERY_DTYPE.itemsize == 68
    assert [m.KP_DTYPE.fields[k][1] for k in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multi_orb_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".cc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle/" not in txt and "import oracle" not in txt and "liborb_oracle" not in txt, f
