"""GPU parity: HIP extractor (through the C ABI) vs the CPU oracle, stage by stage and end to end.  Bit-exact."""
import os
import numpy as np
import pytest

import oracle
from multi_orb_slam_amd import synth

pytestmark = pytest.mark.gpu


def _mk(params, w, h):
    import multi_orb_slam_amd as m
    return m.Extractor(params, w, h)


def _assert_same(kps, desc, okps, odesc):
    assert len(kps) == len(okps), (len(kps), len(okps))
    for f in kps.dtype.names:
        a, b = kps[f], okps[f]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                              b.view(np.uint32) if b.dtype == np.float32 else b), f
    assert np.array_equal(desc, odesc)


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (320, 240, 300), (752, 480, 1200), (641, 479, 500)])
def test_stages_and_end_to_end(w, h, nf):
    import multi_orb_slam_amd as m
    p = m.ExtractorParams(nfeatures=nf)
    ex = _mk([p], w, h)
    img = synth.image(0, 0, w, h)
    kps, desc = ex(img)
    if os.environ.get("MORB_EXPECT_PYRAMID_FORM"):   # (a child of test_resize_chain_form_...: the form it was asked to force did run)
        assert ex.pyramid_form() == int(os.environ["MORB_EXPECT_PYRAMID_FORM"])
    # K1: every pyramid level
    for l, ref in enumerate(oracle.pyramid(img)):
        got = ex.debug_level(0, l)
        assert got.shape == ref.shape and np.array_equal(got, ref), "level %d" % l
        # K2/K3: candidates handed to the quadtree, cell-major order
        cand = ex.debug_candidates(0, l)
        ocand = oracle.cell_candidates(ref)
        assert len(cand) == len(ocand), "level %d" % l
        for f in ("x", "y", "response"):
            assert np.array_equal(cand[f], ocand[f]), (l, f)
    okps, odesc = oracle.extract(img, nfeatures=nf)
    _assert_same(kps, desc, okps, odesc)
    assert len(kps) > nf // 2
    ex.close()


def test_two_cameras_reference_config():
    """configs[0]/[1]: 2 x 640x480, cam 1 with nFeatures, cam 2 with nFeatures/2 (reference src/Tracking.cc:144-145)."""
    import multi_orb_slam_amd as m
    ps = [m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=500)]
    ex = _mk(ps, 640, 480)
    for t in range(3):
        imgs = [synth.image(c, t, 640, 480) for c in range(2)]
        out = ex.extract(imgs)
        for c in range(2):
            okps, odesc = oracle.extract(imgs[c], nfeatures=ps[c].nfeatures)
            _assert_same(out[c][0], out[c][1], okps, odesc)
    ex.close()


def test_mixed_sizes_empty_and_flat_images():
    import multi_orb_slam_amd as m
    ps = [m.ExtractorParams(nfeatures=400), m.ExtractorParams(nfeatures=400), m.ExtractorParams(nfeatures=200)]
    ex = _mk(ps, 640, 480)
    flat = np.full((300, 400), 77, np.uint8)           # no corners at all -> zero keypoints
    imgs = [synth.image(3, 1, 640, 480), flat, None]   # None = empty image: untouched outputs, n = 0
    out = ex.extract(imgs)
    okps, odesc = oracle.extract(imgs[0], nfeatures=400)
    _assert_same(out[0][0], out[0][1], okps, odesc)
    assert len(out[1][0]) == 0 and len(out[2][0]) == 0
    assert len(oracle.extract(flat, nfeatures=400)[0]) == 0
    # a different size on the same handle re-derives the geometry
    imgs = [synth.image(4, 0, 512, 384), synth.image(5, 0, 640, 480), synth.image(6, 0, 320, 240)]
    out = ex.extract(imgs)
    for c in range(3):
        okps, odesc = oracle.extract(imgs[c], nfeatures=ps[c].nfeatures)
        _assert_same(out[c][0], out[c][1], okps, odesc)
    ex.close()


def test_noise_and_low_contrast_images():
    """Noise-only image (dense weak corners -> minTh fallback cells, many candidates) and a low-contrast scene."""
    import multi_orb_slam_amd as m
    rng = synth.hash32(np.arange(640 * 480, dtype=np.uint64) + np.uint64(99))
    noise = (rng % 256).astype(np.uint8).reshape(480, 640)
    low = (128 + (synth.image(2, 0, 640, 480).astype(np.int32) - 128) // 6).astype(np.uint8)
    ex = _mk([m.ExtractorParams(nfeatures=1000)], 640, 480)
    for img in (noise, low):
        kps, desc = ex(img)
        okps, odesc = oracle.extract(img, nfeatures=1000)
        _assert_same(kps, desc, okps, odesc)
    ex.close()


@pytest.mark.parametrize("kind", list(synth.FAMILIES))
@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (1920, 1080, 4000)])
def test_image_families_stage_by_stage(kind, w, h, nf):
    """What the rectangle scenes never show the kernels (round 5, synth.family_image): 1/f-like noise, dithered ramps, soft edges,
    saturated regions, contrast on both sides of the FAST thresholds -- where the roundings of the fixed-point resize and blur decide
    bytes.  Every pyramid level, every level's candidate list and the final keypoints + descriptors against the oracle, at 640x480 (the
    tiled one-launch pyramid) and 1920x1080 (the large-rig form: two tile launches with four pixels per lane), two cameras with different content in one launch."""
    import multi_orb_slam_amd as m
    ex = _mk([m.ExtractorParams(nfeatures=nf)] * 2, w, h)
    imgs = [synth.family_image(kind, c, 1 + c, w, h) for c in range(2)]
    out = ex.extract(imgs)
    for c in range(2):
        for l, ref in enumerate(oracle.pyramid(imgs[c])):
            got = ex.debug_level(c, l)
            assert got.shape == ref.shape and np.array_equal(got, ref), (kind, c, "level %d" % l)
            cand = ex.debug_candidates(c, l)
            ocand = oracle.cell_candidates(ref)
            assert len(cand) == len(ocand), (kind, c, l)
            for f in ("x", "y", "response"):
                assert np.array_equal(cand[f], ocand[f]), (kind, c, l, f)
        okps, odesc = oracle.extract(imgs[c], nfeatures=nf)
        _assert_same(out[c][0], out[c][1], okps, odesc)
        assert len(okps) > nf // 2
    ex.close()


def test_larger_configs():
    """configs[2] (1280x720 @2000) end to end; 1920x1080 @4000 on one camera."""
    import multi_orb_slam_amd as m
    for (w, h, nf) in [(1280, 720, 2000), (1920, 1080, 4000)]:
        ex = _mk([m.ExtractorParams(nfeatures=nf)], w, h)
        img = synth.image(1, 0, w, h)
        kps, desc = ex(img)
        okps, odesc = oracle.extract(img, nfeatures=nf)
        _assert_same(kps, desc, okps, odesc)
        ex.close()


def test_dense_levels_stay_on_the_device_quadtree():
    """1920x1080 @4000: levels 0..5 hold 4500-9400 candidates.  Rounds 1-2 kept 4096 keys in LDS and ran denser levels through
    an HBM-backed second launch, rounds 3-5 held 16 384 keys in LDS; the round-6 quadtree keeps its keys in vector registers
    (up to 40 per thread), so up to 40 959 candidates per level run in the one launch, from the first frame on.  Must equal the oracle;
    a 1080p noise image (> 40 959 candidates on level 0) still goes to the host."""
    import multi_orb_slam_amd as m
    w, h, nf = 1920, 1080, 4000
    ex = _mk([m.ExtractorParams(nfeatures=nf)], w, h)
    for t in range(3):
        img = synth.image(2, t, w, h)
        kps, desc = ex(img)
        okps, odesc = oracle.extract(img, nfeatures=nf)
        _assert_same(kps, desc, okps, odesc)
        assert ex.last_path() == 0, (t, ex.last_path())
    assert max(len(ex.debug_candidates(0, l)) for l in range(8)) > 4096
    rng = synth.hash32(np.arange(w * h, dtype=np.uint64) + np.uint64(5))
    noise = (rng % 256).astype(np.uint8).reshape(h, w)
    kps, desc = ex(noise)
    okps, odesc = oracle.extract(noise, nfeatures=nf)
    _assert_same(kps, desc, okps, odesc)
    assert ex.last_path() == 2 and len(ex.debug_candidates(0, 0)) > 40959
    ex.close()


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (760, 570, 1500), (760, 570, 300)])
def test_register_resident_quadtree_size_classes(w, h, nf):
    """Noise images put 27 703 (640x480) / ~40 000 (760x570) candidates on level 0 and every count down to 161 on the levels above: the
    quadtree's four instantiations (4, 16, 32 and 40 keys per thread; k_octree picks one per (camera, level) from the candidate count)
    all run in one launch, two cameras with different noise, nothing falls back to the host."""
    import multi_orb_slam_amd as m
    ex = _mk([m.ExtractorParams(nfeatures=nf)] * 2, w, h)
    imgs = []
    for c in range(2):
        rng = synth.hash32(np.arange(w * h, dtype=np.uint64) + np.uint64(5 + 977 * c))
        imgs.append((rng % 256).astype(np.uint8).reshape(h, w))
    out = ex.extract(imgs)
    assert ex.last_path() == 0
    n0 = [len(ex.debug_candidates(0, l)) for l in range(8)]
    assert n0[0] > (32768 if w == 760 else 16384) and 16384 < n0[1] <= 32768 and 4096 < n0[3] <= 16384 and n0[7] <= 4096, n0
    for c in range(2):
        okps, odesc = oracle.extract(imgs[c], nfeatures=nf)
        _assert_same(out[c][0], out[c][1], okps, odesc)
    ex.close()


@pytest.mark.parametrize("w,h,nf", [(1920, 1080, 4000), (1280, 720, 2000), (1536, 512, 1500), (1920, 1080, 600), (752, 480, 3000)])
def test_device_quadtree_on_wide_and_dense_levels(w, h, nf):
    """Root strips (16:9 -> 2, 3:1 -> 3 or 4), dense levels (thousands of keys per node in the first passes: the wave-aggregated
    histogram adds), small quotas on dense levels (the careful pass starts early) and large quotas on small images (it never
    starts): every pattern against the oracle's literal std::list restatement, over a few frames and cameras."""
    import multi_orb_slam_amd as m
    ex = _mk([m.ExtractorParams(nfeatures=nf)], w, h)
    for cam, t in ((0, 0), (3, 1), (5, 7)):
        img = synth.image(cam, t, w, h)
        kps, desc = ex(img)
        okps, odesc = oracle.extract(img, nfeatures=nf)
        _assert_same(kps, desc, okps, odesc)
        assert ex.last_path() == 0
    ex.close()


def test_levels_beyond_the_device_limit_fall_back_to_the_host_quadtree():
    """MORB_OCT_MAX_KEYS lowers the device quadtree's candidate limit: the levels beyond it are flagged by the kernel and the
    run is redone on the host path -- same results."""
    import os
    import multi_orb_slam_amd as m
    os.environ["MORB_OCT_MAX_KEYS"] = "500"
    try:
        ex = _mk([m.ExtractorParams(nfeatures=1000)] * 2, 640, 480)
    finally:
        os.environ.pop("MORB_OCT_MAX_KEYS")
    imgs = [synth.image(c, 3, 640, 480) for c in range(2)]
    out = ex.extract(imgs)
    assert ex.last_path() == 2 and len(ex.debug_candidates(0, 0)) > 500
    for c in range(2):
        okps, odesc = oracle.extract(imgs[c], nfeatures=1000)
        _assert_same(out[c][0], out[c][1], okps, odesc)
    ex.close()


def test_resident_path_and_determinism():
    import multi_orb_slam_amd as m
    ex = _mk([m.ExtractorParams(nfeatures=1000)] * 2, 640, 480)
    imgs = [synth.image(c, 5, 640, 480) for c in range(2)]
    for c in range(2):
        ex.upload(c, imgs[c])
    ex.run()
    first = [ex.download(c) for c in range(2)]
    ex.run()  # same resident images again: identical output
    second = [ex.download(c) for c in range(2)]
    for c in range(2):
        assert np.array_equal(first[c][0], second[c][0]) and np.array_equal(first[c][1], second[c][1])
        okps, odesc = oracle.extract(imgs[c], nfeatures=1000)
        _assert_same(first[c][0], first[c][1], okps, odesc)
    ex.close()


def test_host_quadtree_path_equals_device_quadtree_path():
    """MORB_HOST_OCTREE=1 forces the host quadtree (also the fallback when a level exceeds the device limits)."""
    import os
    import multi_orb_slam_amd as m
    imgs = [synth.image(c, 2, 640, 480) for c in range(2)]
    outs = []
    for env in ("0", "1"):
        os.environ["MORB_HOST_OCTREE"] = env
        ex = _mk([m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=500)], 640, 480)
        outs.append(ex.extract(imgs))
        ex.close()
    os.environ.pop("MORB_HOST_OCTREE")
    for c in range(2):
        assert outs[0][c][0].tobytes() == outs[1][c][0].tobytes() and np.array_equal(outs[0][c][1], outs[1][c][1])
        okps, odesc = oracle.extract(imgs[c], nfeatures=(1000, 500)[c])
        _assert_same(outs[0][c][0], outs[0][c][1], okps, odesc)


@pytest.mark.parametrize("nf", [50, 217, 1500, 3000])
def test_quota_sweep_device_quadtree(nf):
    """Different quotas drive the quadtree through different full/careful pass patterns."""
    import multi_orb_slam_amd as m
    ex = _mk([m.ExtractorParams(nfeatures=nf)], 640, 480)
    for t in range(2):
        img = synth.image(7, t, 640, 480)
        kps, desc = ex(img)
        okps, odesc = oracle.extract(img, nfeatures=nf)
        _assert_same(kps, desc, okps, odesc)
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kw", [(640, 480, dict(nfeatures=1000, ini_th_fast=12, min_th_fast=3)),
                                    (640, 480, dict(nfeatures=500, ini_th_fast=40, min_th_fast=20)),
                                    (700, 333, dict(nfeatures=800, scale_factor=1.1, nlevels=10)),
                                    (512, 512, dict(nfeatures=1500, scale_factor=1.5, nlevels=4)),
                                    (777, 333, dict(nfeatures=600, scale_factor=1.3, nlevels=5, ini_th_fast=15, min_th_fast=7)),
                                    (160, 120, dict(nfeatures=100, nlevels=3))])
def test_odd_parameter_sets_equal_the_oracle(w, h, kw):
    """Other FAST thresholds (the packed quick test's T), level counts, level steps (pyramid tables, quotas) and image shapes:
    keypoint records and descriptors bit for bit."""
    import multi_orb_slam_amd as m
    p = m.ExtractorParams(**kw)
    ex = _mk([p, p], w, h)
    imgs = [synth.image(7 + c, 1, w, h) for c in range(2)]
    out = ex.extract(imgs)
    for c in range(2):
        ok, od = oracle.extract(imgs[c], nfeatures=p.nfeatures, scale_factor=p.scale_factor, nlevels=p.nlevels,
                                ini_th=p.ini_th_fast, min_th=p.min_th_fast)
        assert out[c][0].tobytes() == ok.tobytes() and np.array_equal(out[c][1], od)
    assert ex.last_path() == 0


@pytest.mark.gpu
def test_a_level_the_reference_leaves_undefined_still_runs():
    """333 x 777: round(width / height) == 0 quadtree roots in the reference (undefined, the oracle refuses); the product takes one
    root and returns features -- no parity is claimed, the call must simply succeed."""
    import multi_orb_slam_amd as m
    ex = _mk([m.ExtractorParams(nfeatures=600)], 333, 777)
    out = ex.extract([synth.image(3, 0, 333, 777)])
    assert 300 < len(out[0][0]) <= 600 + 3 * 8 and out[0][1].shape == (len(out[0][0]), 32)
    with pytest.raises(ValueError):
        oracle.extract(synth.image(3, 0, 333, 777), nfeatures=600)


# The forms a pyramid can take besides the one-launch tile kernel of small rigs, forced at the sizes of this file's stage tests (the
# arrangement is chosen once per process, so each form runs them in a child): the generic chain (one k_resize launch per level: what odd
# parameter sets fall back to), the large-rig tile launches (k_pyramid_tiled4: levels 1..3 below level 0 and 4.. below level 3 in
# 128 x 64 tiles), the same with tiles so small that every image is dozens of them with ragged last rows and columns, and with the split
# after level 1 / level 5 / beyond the last level (one launch does everything).  MORB_TEST_PYRAMID_PLAN = "tile_w,tile_h,split" is read
# by the TESTS (tests/conftest.py) and handed to orbx_debug_pyramid_plan -- the library itself has no such switch.
PYRAMID_FORMS = {
    "generic_chain": (2, {"MORB_PYR_CHAIN": "2"}),
    "tiled4": (3, {}),
    "tiled4_small_tiles": (3, {"MORB_TEST_PYRAMID_PLAN": "32,16,0"}),
    "tiled4_wide_tiles_split1": (3, {"MORB_TEST_PYRAMID_PLAN": "256,24,1"}),
    "tiled4_split5": (3, {"MORB_TEST_PYRAMID_PLAN": "64,64,5"}),
    "tiled4_one_launch": (3, {"MORB_TEST_PYRAMID_PLAN": "64,32,99"}),
}


@pytest.mark.parametrize("form", sorted(PYRAMID_FORMS))
def test_resize_chain_form_equals_the_oracle_at_every_size(form):
    """Large rigs build their pyramid with four pixels per lane from a per-group table of byte selectors and coefficient pairs instead of
    the one-launch tile kernel of small rigs, parameter sets outside both take one plain launch per level; the stage tests of this file
    -- every level byte for byte, odd sizes included -- run once more in a child with each form forced at their sizes
    (MORB_EXPECT_PYRAMID_FORM makes the child check which one ran -- where the parameter set admits the form at all)."""
    import subprocess, sys
    want, extra = PYRAMID_FORMS[form]
    env = dict(os.environ, MORB_PYR_CHAIN="1", MORB_EXPECT_PYRAMID_FORM=str(want)); env.update(extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "stages_and_end_to_end or mixed_sizes or larger_configs or odd_parameter or image_families"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-2000:]
    assert r.returncode == 0 and " passed" in tail, tail



@pytest.mark.parametrize("env", [{}, {"MORB_FAST_FORM": "2", "MORB_PYR_CHAIN": "1"}], ids=["default_forms", "large_rig_forms"])
def test_random_parameter_sets_equal_the_oracle(env):
    """Twelve random parameter sets (image size, scale factor, level count, FAST thresholds on both sides of the packed quick test's 127,
    feature count, image family) through the whole extraction, in the small-rig and in the large-rig forms of the kernels
    (tools/fuzz_extractor_random.py; the same tool runs hundreds of cases by hand)."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_extractor_random.py")
    r = subprocess.run([sys.executable, tool, "12", "5"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and "ok: " in out and "differs" not in out, out
