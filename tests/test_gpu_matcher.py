"""GPU parity: HIP matcher (through the C ABI) vs the CPU oracle.  Bit-exact for every output."""
import os
import numpy as np
import pytest

import oracle
import helpers
from multi_orb_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    import multi_orb_slam_amd as m
    mt = m.Matcher(0.8, True)
    yield mt
    mt.close()


# nq >= 64 and nr >= 64 take the matrix-core kernel (ragged last tile, several slices, ragged query blocks), the rest
# the one-query-per-lane kernel
@pytest.mark.parametrize("nq,nr", [(1, 1), (1, 17), (63, 15), (64, 64), (65, 1000), (1000, 1000), (257, 4097),
                                   (1500, 500), (5, 0), (0, 5), (64, 65), (300, 127), (2000, 3001), (513, 20000)])
def test_top2_bit_exact(matcher, nq, nr):
    r = synth.descriptors(max(nr, 1), 42)[:nr]
    q = synth.perturbed_queries(synth.descriptors(max(nq, 1), 42), 7)[:nq] if nq else np.zeros((0, 32), np.uint8)
    bi, bd, sd = matcher.hamming_top2(q, r)
    obi, obd, osd = oracle.bf_top2(q, r)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)


@pytest.mark.parametrize("nq,nr,nbase,maxflip,seed", [(3000, 9000, 2, 1, 1), (257, 4097, 1, 0, 2), (1000, 1000, 5, 3, 3), (6000, 129, 40, 1, 4)])
def test_top2_on_tie_heavy_rows_in_every_form(matcher, nq, nr, nbase, maxflip, seed):
    """Rows drawn from a handful of base vectors with 0-3 flipped bits: equal distances everywhere, exact duplicates, best == second,
    a few complemented rows (distance 256).  The matrix-core forms only do their key work for quarter-blocks in which some key is
    below a lane's `second` (mt_step): on such data nearly every key TIES with it -- which must neither be skipped wrongly nor
    change which duplicate is reported first.  FP4, int8 and popcount forms against the oracle's brute force."""
    import multi_orb_slam_amd as m
    rng = np.random.RandomState(seed)

    def rows(n):
        base = rng.randint(0, 256, (nbase, 32)).astype(np.uint8)
        d = base[rng.randint(0, nbase, n)].copy()
        for i in range(n):
            for _ in range(rng.randint(0, maxflip + 1)):
                b = rng.randint(0, 256); d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        far = rng.rand(n) < 0.02
        d[far] = ~d[far]
        return np.ascontiguousarray(d)
    q = rows(nq); r = rows(nr)
    r[:min(nq, nr) // 2] = q[:min(nq, nr) // 2]
    e = oracle.bf_top2(q, r)
    for mc, fp4 in ((1, -1), (1, 0), (0, -1)):
        pm = m.Matcher.use_matrix_cores(mc); pf = m.Matcher.use_fp4_top2(fp4)
        try:
            g = matcher.hamming_top2(q, r)
        finally:
            m.Matcher.use_matrix_cores(pm); m.Matcher.use_fp4_top2(pf)
        assert all(np.array_equal(a, b) for a, b in zip(g, e)), (mc, fp4)


def test_top2_ties_and_duplicates(matcher):
    # exact duplicates: the first index wins, the duplicate becomes the second best (App. A-9)
    r = synth.descriptors(300, 3)
    r[200] = r[10]; r[299] = r[10]; r[150] = r[149]
    q = np.concatenate([r[[10, 149, 200]], ~r[[5]], np.zeros((1, 32), np.uint8)])
    bi, bd, sd = matcher.hamming_top2(q, r)
    obi, obd, osd = oracle.bf_top2(q, r)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert bi[0] == 10 and bd[0] == 0 and sd[0] == 0 and bi[2] == 10
    # all-ones vs all-zero: distance 256 never registers (initial value 256, strict '<')
    z = np.zeros((4, 32), np.uint8); o = np.full((3, 32), 255, np.uint8)
    bi, bd, sd = matcher.hamming_top2(o, z)
    assert (bi == -1).all() and (bd == 256).all() and (sd == 256).all()
    # the same on the matrix-core path, with duplicates falling into different tiles, half-waves and slices
    r = synth.descriptors(5000, 8)
    r[4999] = r[3]; r[70] = r[3]; r[36] = r[35]; r[2047] = r[2048]
    q = np.concatenate([r[[3, 35, 2048, 4999]], synth.perturbed_queries(r[:96], 5)])
    bi, bd, sd = matcher.hamming_top2(q, r)
    obi, obd, osd = oracle.bf_top2(q, r)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert list(bi[:4]) == [3, 35, 2047, 3] and (sd[:4] == 0).all()
    z = np.zeros((100, 32), np.uint8); o = np.full((130, 32), 255, np.uint8)
    bi, bd, sd = matcher.hamming_top2(o, z)
    assert (bi == -1).all() and (bd == 256).all() and (sd == 256).all()
    z[77, 0] = 1; z[99, 0] = 3
    bi, bd, sd = matcher.hamming_top2(o, z)
    assert (bi == 99).all() and (bd == 254).all() and (sd == 255).all()


# sizes with nq >= 64 and nr % 8 == 0 take the matrix-core kernel (ragged query blocks, ragged and partial reference tiles,
# several tiles per workgroup), the others the xor/popcount kernel
@pytest.mark.parametrize("nq,nr", [(1, 1), (3, 8), (70, 513), (129, 1024), (200, 1000), (64, 4099), (64, 64), (300, 72),
                                   (257, 2056), (1000, 1000), (513, 8200)])
def test_matrix_bit_exact(matcher, nq, nr):
    r = synth.descriptors(nr, 9); q = synth.descriptors(nq, 10)
    q[0] = r[0]
    out = matcher.hamming_matrix(q, r)
    assert np.array_equal(out, oracle.hamming_matrix(q, r))


def test_matrix_extremes(matcher):
    # all bits different = 256 (the int8 dot product is -256), identical = 0, on the matrix-core path
    z = np.zeros((96, 32), np.uint8); o = np.full((128, 32), 255, np.uint8)
    assert (matcher.hamming_matrix(z, o) == 256).all() and (matcher.hamming_matrix(o, o) == 0).all()
    one = z.copy(); one[:, 31] = 0x80
    assert (matcher.hamming_matrix(one, np.concatenate([z, o])) == np.r_[np.full(96, 1), np.full(128, 255)][None, :]).all()


def test_matrix_core_and_popcount_forms_agree(matcher):
    # orbm_use_matrix_cores switches the two all-pairs kernels between their forms at run time: same bits either way
    import multi_orb_slam_amd as m
    r = synth.descriptors(3000, 21); q = synth.perturbed_queries(synth.descriptors(700, 21), 4)
    out = {}
    try:
        for on in (1, 0):
            m.Matcher.use_matrix_cores(on)
            out[on] = (matcher.hamming_matrix(q, r[:2048]),) + tuple(matcher.hamming_top2(q, r))
        # the matrix-core top-2 once more with int8 instead of FP4 arithmetic (orbm_use_fp4_top2)
        m.Matcher.use_matrix_cores(1)
        assert m.Matcher.use_fp4_top2(0) == -1
        out[2] = (out[1][0],) + tuple(matcher.hamming_top2(q, r))
    finally:
        assert m.Matcher.use_matrix_cores(-1) == 1
        assert m.Matcher.use_fp4_top2(-1) == 0
    for a, b, c in zip(out[0], out[1], out[2]):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.array_equal(out[1][0], oracle.hamming_matrix(q, r[:2048]))
    ebi, ebd, esd = oracle.bf_top2(q, r)
    assert np.array_equal(out[1][1], ebi) and np.array_equal(out[1][2], ebd) and np.array_equal(out[1][3], esd)


def test_matrix_properties_full_size(matcher):
    # size-independent properties at an all-pairs size: symmetry, zero diagonal, row checksum vs popcount identity
    n = 4000
    d = synth.descriptors(n, 123)
    m = matcher.hamming_matrix(d, d)
    assert (np.diag(m) == 0).all() and np.array_equal(m, m.T)
    rows = np.array([0, 1, 1999, 3999])
    assert np.array_equal(m[rows], oracle.hamming_matrix(d[rows], d))
    bi, bd, sd = matcher.hamming_top2(d, d)
    assert np.array_equal(bi, np.arange(n)) and (bd == 0).all()
    mm = m.astype(np.int32); np.fill_diagonal(mm, 1000)
    assert np.array_equal(sd, np.minimum(mm.min(1), 256))


@pytest.mark.parametrize("n_per_cam,seed", [([400, 300], 1), ([1500, 700], 2), ([50], 3), ([0, 20], 4)])
def test_grid_and_area_queries(matcher, n_per_cam, seed):
    fr = helpers.make_frame_arrays(n_per_cam, 640, 480, seed)
    import multi_orb_slam_amd as m
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    cs, items = F.grid(); ocs, oitems = oracle.grid_csr(OF)
    assert np.array_equal(cs, ocs) and np.array_equal(items, oitems)
    pts = [(320.0, 240.0, 40.0, -1, -1), (0.5, 0.5, 25.0, 0, 3), (639.0, 479.0, 60.0, 2, -1), (-30.0, 100.0, 20.0, -1, -1),
           (700.0, 100.0, 70.0, 1, 2), (100.3, 200.7, 10.0, 0, 0), (333.0, 111.0, 300.0, -1, -1)]
    for cam in range(len(n_per_cam)):
        for (x, y, r, lo, hi) in pts:
            got = matcher.features_in_area(F, cam, x, y, r, lo, hi)
            exp = oracle.features_in_area(OF, cam, x, y, r, lo, hi)
            assert np.array_equal(got, exp), (cam, x, y, r, lo, hi)
    F.close()


@pytest.mark.parametrize("n_per_cam,resident,seed", [([400, 300], (True, True), 1), ([1500, 700], (True, False), 2), ([50], (True,), 3),
                                                     ([0, 20], (False, True), 4), ([1000, 1000, 600, 900], (False, True, True, False), 5),
                                                     ([3000, 3000, 3000], (True, True, True), 6), ([300, 200], (False, False), 7)])
def test_frame_created_with_resident_descriptors_equals_the_host_built_frame(matcher, n_per_cam, resident, seed):
    """orbm_frame_create_resident (round 4: what the C++ ORBmatcher uploads frames through) -- per-feature fields in one staging
    block, the 64x48 grid built by k_frame_build_small on the device, the descriptor rows of some cameras read from DEVICE memory
    where an extractor left them -- must give the frame orbm_frame_create gives: the same grid (cell starts, item order), the
    same candidates for every window, the same search results; with every mix of resident / host cameras, an empty camera, a
    frame beyond 8192 features (which takes the host-built path) and a mapping that is NOT camera-major."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import rt
    fr = helpers.make_frame_arrays(n_per_cam, 640, 480, seed)
    if seed == 7:      # global order interleaves the cameras: cam_of / local_of say so, the descriptors stay per camera
        n = len(fr["un_x"])
        perm = np.random.RandomState(3).permutation(n)
        for k in ("un_x", "un_y", "octave", "angle", "uright", "cam_of", "local_of"):
            fr[k] = np.ascontiguousarray(fr[k][perm])
    data = m.FrameData(**fr); OF = oracle.FrameData(**fr)
    bufs = []
    ptrs = []
    for c, d in enumerate(data.descs):
        if resident[c] and len(d):
            b = rt.DeviceBuffer(max(d.nbytes, 32)); b.upload(d); bufs.append(b); ptrs.append(b.ptr)
        else:
            ptrs.append(0)
    rt.device_sync()
    F = matcher.frame(data, resident=ptrs)
    G = matcher.frame(data)
    cs, items = F.grid(); gcs, gitems = G.grid(); ocs, oitems = oracle.grid_csr(OF)
    assert np.array_equal(cs, ocs) and np.array_equal(items, oitems)
    assert np.array_equal(cs, gcs) and np.array_equal(items, gitems)
    nq = min(1500, max(10, len(fr["un_x"])))
    q = helpers.make_queries(fr, nq, seed + 70, th=20.0, blocks=1)
    for check_ori in (True, False):
        matcher.check_orientation = check_ori
        n1, mo1 = matcher.SearchByProjection(F, q)
        n2, mo2 = matcher.SearchByProjection(G, q)
        on, omo = oracle.search_by_projection_frames(OF, q, 100, check_ori)
        assert n1 == on and np.array_equal(mo1, omo)
        assert n2 == on and np.array_equal(mo2, omo)
    matcher.check_orientation = True
    idx, dist, cnt = matcher.project_candidates(F, q[:64], 512)
    idx2, dist2, cnt2 = matcher.project_candidates(G, q[:64], 512)
    assert np.array_equal(cnt, cnt2) and np.array_equal(idx, idx2) and np.array_equal(dist, dist2)
    F.close(); G.close()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("n_per_cam,nq,th,blocks,seed", [([1000, 500], 1200, 15.0, 1, 1), ([1000, 1000], 2500, 30.0, 1, 2),
                                                        ([300, 200], 900, 15.0, 2, 3), ([2000, 2000], 3000, 15.0, 0, 4),
                                                        ([64], 10, 7.0, 1, 5), ([1000, 1000], 2000, 30.0, 1, 6),
                                                        ([1000, 1000], 4000, 30.0, 1, 7)])
def test_search_by_projection_frames(matcher, n_per_cam, nq, th, blocks, seed):
    import multi_orb_slam_amd as m
    fr = helpers.make_frame_arrays(n_per_cam, 640, 480, seed)
    q = helpers.make_queries(fr, nq, seed + 40, th=th, blocks=blocks)
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    for check_ori in (True, False):
        matcher.check_orientation = check_ori
        n, mo = matcher.SearchByProjection(F, q)
        on, omo = oracle.search_by_projection_frames(OF, q, 100, check_ori)
        assert n == on and np.array_equal(mo, omo)
        assert n > 0
    matcher.check_orientation = True
    # ordered candidate lists themselves
    nchk = min(200, nq)
    idx, dist, cnt = matcher.project_candidates(F, q[:nchk], 512)
    for i in range(0, nchk, 7):
        # oracle candidate order + the right-coordinate gate
        cand = oracle.features_in_area(OF, int(q["cam"][i]), float(q["u"][i]), float(q["v"][i]), float(q["radius"][i]),
                                       int(q["min_level"][i]), int(q["max_level"][i]))
        ur = fr["uright"][cand]
        keep = ~((ur > 0) & (np.abs(np.float32(q["ur"][i]) - ur) > q["radius"][i]))
        cand = cand[keep]
        assert cnt[i] == len(cand) and np.array_equal(idx[i, :cnt[i]], cand)
        alld = np.concatenate(fr["descs"])
        exp = [oracle.descriptor_distance(q["desc"][i], alld[g]) for g in cand]
        assert np.array_equal(dist[i, :cnt[i]], np.array(exp, np.uint16))
    F.close()


@pytest.mark.parametrize("n,nq,seed", [(1000, 800, 1), (1500, 2000, 2)])
def test_search_by_projection_points(matcher, n, nq, seed):
    import multi_orb_slam_amd as m
    fr = helpers.make_frame_arrays([n, n // 2], 640, 480, seed)
    q = helpers.make_queries(fr, nq, seed + 9, th=3 * 4.0)
    q["cam"] = 0
    q["max_level"] = np.maximum(q["max_level"], 0); q["min_level"] = q["max_level"] - 1
    occ = (helpers.rand_unit(n + n // 2, seed + 77) < 0.2).astype(np.uint8)
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    for ratio in (0.8, 0.6):
        matcher.nnratio = ratio
        for o in (None, occ):
            cnt, mo = matcher.SearchByProjectionPoints(F, q, o)
            ocnt, omo = oracle.search_by_projection_points(OF, q, o, ratio, 100)
            assert cnt == ocnt and np.array_equal(mo, omo)
            assert cnt > 0
    F.close()


def test_no_device_fallback_is_an_error():
    import multi_orb_slam_amd as m
    with pytest.raises(m.OrbError):
        m.Matcher(device=99)


# ------------------------------------------------------------------------------------------------ device-resident frame
def _depth_image(w, h, seed):
    d = 0.5 + (helpers.rand_unit(w * h, seed) * 8).astype(np.float32).reshape(h, w)
    d[helpers.rand_unit(w * h, seed + 1).reshape(h, w) < 0.15] = 0.0      # holes: no depth -> uRight = -1
    return d.astype(np.float32)


@pytest.mark.parametrize("sizes", [[(640, 480, 1000), (640, 480, 500)], [(320, 240, 300)],
                                   [(640, 480, 700), (640, 480, 300), (640, 480, 900), (640, 480, 40), (640, 480, 500)]])
def test_frame_from_device_matches_host_assembly(matcher, sizes):
    """Merge + ComputeStereoFromRGBD + AssignFeaturesToGrid on the device == oracle on the same keypoints.  (Five cameras: the large-rig
    assembly -- k_frame_fill + one k_grid_cam workgroup per camera -- instead of the single-workgroup build of rigs up to four.)"""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import rt
    W, H = sizes[0][0], sizes[0][1]
    ex = m.Extractor([m.ExtractorParams(nfeatures=nf) for _, _, nf in sizes], W, H)
    imgs = [synth.image(c, 0, W, H) for c in range(len(sizes))]
    per_cam = ex.extract(imgs)
    depths = [_depth_image(W, H, 50 + c) for c in range(len(sizes))]
    dbufs = []
    for d in depths:
        b = rt.DeviceBuffer(d.nbytes); b.upload(d); dbufs.append(b)
    matcher.set_stream(ex.stream)
    cams = [(ex.device_keypoints(c), ex.device_descriptors(c), len(per_cam[c][0]), dbufs[c].ptr, W) for c in range(len(sizes))]
    F = matcher.frame_from_device(cams, 40.0, (0, 0, W, H))
    k, d, ur, dp = F.download()
    ek = np.concatenate([p[0] for p in per_cam]); ed = np.concatenate([p[1] for p in per_cam])
    assert k.tobytes() == ek.tobytes() and np.array_equal(d, ed)
    eur = []; edp = []
    for c in range(len(sizes)):
        a, b = oracle.stereo_from_depth(per_cam[c][0], depths[c], 40.0)
        eur.append(a); edp.append(b)
    eur = np.concatenate(eur); edp = np.concatenate(edp)
    assert np.array_equal(ur.view(np.uint32), eur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), edp.view(np.uint32))
    assert (ur > 0).sum() > len(ur) // 2 and (ur < 0).sum() > 10
    # grid
    cam_of = np.concatenate([np.full(len(p[0]), c, np.int32) for c, p in enumerate(per_cam)])
    loc = np.concatenate([np.arange(len(p[0]), dtype=np.int32) for p in per_cam])
    OF = oracle.FrameData(ek["x"], ek["y"], ek["octave"], ek["angle"], eur, cam_of, loc, [p[1] for p in per_cam], (0, 0, W, H))
    cs, items = F.grid(); ocs, oitems = oracle.grid_csr(OF)
    assert np.array_equal(cs, ocs) and np.array_equal(items, oitems)
    # searches on the device-built frame
    fr = dict(un_x=ek["x"], un_y=ek["y"], octave=ek["octave"], angle=ek["angle"], uright=eur, cam_of=cam_of, local_of=loc,
              descs=[p[1] for p in per_cam], bounds=(0.0, 0.0, float(W), float(H)))
    q = helpers.make_queries(fr, 1500, 3, th=15.0)
    n, mo = matcher.SearchByProjection(F, q)
    on, omo = oracle.search_by_projection_frames(OF, q, 100, True)
    assert n == on and np.array_equal(mo, omo) and n > 100
    # cross-camera top-2 in one launch
    bi, bd, sd = matcher.cross_top2(F)
    off = 0
    for c, p in enumerate(per_cam):
        others = [pp[1] for o, pp in enumerate(per_cam) if o != c]
        refs = np.concatenate(others) if others else np.zeros((0, 32), np.uint8)
        ebi, ebd, esd = oracle.bf_top2(p[1], refs)
        nc = len(p[0])
        assert np.array_equal(bi[off:off + nc], ebi) and np.array_equal(bd[off:off + nc], ebd) and np.array_equal(sd[off:off + nc], esd)
        off += nc
    F.close(); matcher.set_stream(None); ex.close()


def test_device_resolve_long_dependency_chains_and_host_fallback_agree():
    """Adversarial first-come chains: many identical queries fight over the same few features, so the Jacobi sweeps of
    the device resolve have to propagate claims query by query.  Device resolve, host resolve (MORB_HOST_RESOLVE=1)
    and the oracle must agree exactly."""
    import os
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd._lib import QUERY_DTYPE
    n = 400
    fr = helpers.make_frame_arrays([n], 640, 480, 9, with_right=False)
    fr["un_x"] = (100 + (np.arange(n) % 20) * 2).astype(np.float32); fr["un_y"] = (100 + (np.arange(n) // 20) * 2).astype(np.float32)
    fr["octave"][:] = 0
    fr["descs"][0][:] = synth.descriptors(1, 5)[0]            # all features identical: pure order decides
    nq = 600
    q = np.zeros(nq, QUERY_DTYPE)
    q["u"] = 120; q["v"] = 120; q["radius"] = 60; q["ur"] = -1; q["min_level"] = -1; q["max_level"] = -1; q["cam"] = 0
    q["blocks"] = 1; q["angle"] = fr["angle"][0]; q["desc"] = synth.descriptors(1, 5)[0]
    q["blocks"][::7] = 0                                     # a few non-blocking claims that later queries overwrite
    OF = oracle.FrameData(**fr)
    results = []
    for env in ("0", "1"):
        os.environ["MORB_HOST_RESOLVE"] = env
        mt = m.Matcher(0.8, False)
        F = mt.frame(m.FrameData(**fr))
        results.append(mt.SearchByProjection(F, q))
        cntp, mop = mt.SearchByProjectionPoints(F, q)
        results.append((cntp, mop))
        F.close(); mt.close()
    os.environ.pop("MORB_HOST_RESOLVE")
    on, omo = oracle.search_by_projection_frames(OF, q, 100, False)
    onp, omop = oracle.search_by_projection_points(OF, q, None, 0.8, 100)
    for k, (cnt, mo) in enumerate(results):
        e_n, e_m = (on, omo) if k % 2 == 0 else (onp, omop)
        assert cnt == e_n and np.array_equal(mo, e_m), k
    assert on >= n                                           # every feature ends up claimed


def test_frame_search_resolves_the_same_as_jacobi_sweeps_and_as_the_monotone_iteration():
    """SearchByProjection(Frame, Frame) resolves with k_resolve_mono by default and with k_resolve's Jacobi sweeps under
    MORB_RESOLVE_MONO=0 (read once per process: a child process runs the frame-search tests of this file, adversarial claim
    chains included, on the other kernel).  Both must equal the oracle -- this process has just shown it for the default."""
    import subprocess, sys
    env = dict(os.environ, MORB_RESOLVE_MONO="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "search_by_projection_frames or long_dependency_chains"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-2000:]
    assert r.returncode == 0 and " passed" in tail, tail


def _large_frame_case(kind):
    """(frame arrays, queries, occupied) of the large-frame searches below.  8 x 4000 features: "mixed" 20000 queries in random
    camera order, "own" 32000 camera-major queries (the motion stream's shape: 4000 per camera), "occupied" with a third of the
    features taken and non-blocking queries among the rest, "invalid" with queries that name no camera of the frame.  2 x 9000
    features: "wide" 7000 queries (72 KB of tables per camera), "crowded" 24000 queries (more than 4096 per camera: beyond the
    per-camera launch)."""
    two = kind in ("wide", "crowded")
    fr = helpers.make_frame_arrays([9000] * 2 if two else [4000] * 8, 1920, 1080, 17)
    occ = None
    if kind == "mixed":
        q = helpers.make_queries(fr, 20000, 23, th=30.0)
    elif kind == "own":
        q = helpers.make_queries(fr, 32000, 29, th=25.0)
        order = np.argsort(q["cam"], kind="stable")
        q = np.ascontiguousarray(q[order])
    elif kind == "wide":
        q = helpers.make_queries(fr, 7000, 47, th=25.0)
    elif kind == "crowded":
        q = helpers.make_queries(fr, 24000, 31, th=20.0)
    elif kind == "occupied":
        q = helpers.make_queries(fr, 24000, 37, th=30.0, blocks=2)
        occ = (helpers.rand_unit(32000, 41) < 0.33).astype(np.uint8)
    else:
        q = helpers.make_queries(fr, 16000, 43, th=30.0)
        q["cam"][::7] = 9
        q["cam"][3::11] = -1
    return fr, q, occ


@pytest.mark.parametrize("kind", ["mixed", "own", "wide", "crowded", "occupied", "invalid"])
def test_projection_search_large_frame_resolves_per_camera(matcher, kind):
    """8 cameras x 4000 features (configs[4] scale): the claim tables of the whole frame do not fit one workgroup's LDS.  A frame
    search never crosses cameras, so every camera gets a workgroup of its own with that camera's tables in LDS (k_rs_mono_cam, one
    launch); more than 4096 queries on one camera ("crowded") keep the tables in HBM and one launch per sweep.  Either way the
    result is the oracle's sequential loop, bit for bit, and the search stays on the device."""
    import multi_orb_slam_amd as m
    fr, q, occ = _large_frame_case(kind)
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    n, mo = matcher.SearchByProjection(F, q, occupied=occ)
    qo = q
    if kind == "invalid":   # (a camera the frame does not have is an out-of-bounds read in the reference and in the oracle: the
        qo = q.copy()       # library treats such a point as one without candidates -- stated to the oracle as a window far outside)
        bad = (q["cam"] < 0) | (q["cam"] >= 8)
        assert bad.sum() > 3000
        qo["cam"][bad] = 0; qo["u"][bad] = -1.0e6
    on, omo = oracle.search_by_projection_frames(OF, qo, 100, True, occ)
    assert n == on and np.array_equal(mo, omo) and n > 3000
    status, nm, sweeps, longest = matcher.last_resolve()
    assert status == 0 and nm == n and sweeps >= 2          # resolved on the device (the host path leaves no sweep count)
    F.close()


@pytest.mark.parametrize("n_per_cam,nq,seed", [([17000, 0, 3000], 4500, 3), ([6000, 1, 9000, 5000], 8000, 5), ([2500] * 8, 2500, 7),
                                             ([19000, 500], 6000, 9)])
def test_large_frame_unbalanced_rigs(matcher, n_per_cam, nq, seed):
    """The per-camera resolve sizes its LDS for the LARGEST camera: 17 000 features on one camera (144 KB of tables) next to an empty
    one, a one-feature camera, a rig whose cameras all fit easily; 19 000 on one camera is beyond a workgroup and keeps the per-sweep
    form.  All equal the oracle."""
    import multi_orb_slam_amd as m
    fr = helpers.make_frame_arrays(n_per_cam, 1920, 1080, seed)
    q = helpers.make_queries(fr, nq, seed + 1, th=25.0, blocks=2)
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    n, mo = matcher.SearchByProjection(F, q)
    on, omo = oracle.search_by_projection_frames(OF, q, 100, True)
    assert n == on and np.array_equal(mo, omo) and n > nq // 10
    assert matcher.last_resolve()[0] == 0
    F.close()


def test_large_frame_per_sweep_form_is_still_exact():
    """MORB_RS_PER_CAMERA=0 sends every large frame to the per-sweep form (tables in HBM): the cases above must equal the oracle on
    it too (the switch is read once per process, hence the child)."""
    import subprocess, sys
    env = dict(os.environ, MORB_RS_PER_CAMERA="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "large_frame_resolves_per_camera or large_frame_unbalanced"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-2000:]
    assert r.returncode == 0 and " passed" in tail, tail


@pytest.mark.parametrize("world,cams_per_rank", [(1, 2), (3, 2), (4, 1)])
def test_cross_top2_from_a_gathered_buffer(world, cams_per_rank):
    """orbm_cross_top2_gathered: what every rank does after the one all-gather of a timestep.  The gathered buffer is
    built by hand here: per rank cap_rows descriptor rows (its cameras packed back to back, unused rows = garbage) and
    the count trailer.  Every rank's answer must equal brute force against the other cameras in global order."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import rt
    cap_rows = 700
    block_bytes = cap_rows * 32 + 256
    n_cams = world * cams_per_rank
    counts = [(150 + 37 * g) % 301 + (0 if g != 1 else -((150 + 37) % 301)) for g in range(n_cams)]   # camera 1 is empty
    base = synth.descriptors(400, 21)
    descs = []
    for g in range(n_cams):
        d = synth.perturbed_queries(base, seed=50 + g, flip_p=0.05)[:counts[g]] if counts[g] else np.zeros((0, 32), np.uint8)
        descs.append(np.ascontiguousarray(d))
    from multi_orb_slam_amd.dist import pack_export_block, unpack_gathered, BLOCK_TRAILER
    assert block_bytes == cap_rows * 32 + BLOCK_TRAILER
    buf = np.concatenate([pack_export_block([descs[r * cams_per_rank + c] for c in range(cams_per_rank)], cap_rows) for r in range(world)])
    back = unpack_gathered(buf, world, cap_rows, cams_per_rank)          # the numpy restatement the gloo test relies on
    assert all(np.array_equal(back[g], descs[g]) for g in range(n_cams))
    dev = rt.DeviceBuffer(buf.nbytes); dev.upload(buf)
    matcher = m.Matcher()
    for rank in range(world):
        bi, bd, sd, cnt = matcher.cross_top2_gathered(dev.ptr, world, block_bytes, cap_rows, cams_per_rank, rank)
        assert cnt == counts
        off = 0
        for c in range(cams_per_rank):
            g = rank * cams_per_rank + c
            others = [descs[o] for o in range(n_cams) if o != g]
            refs = np.concatenate(others) if others else np.zeros((0, 32), np.uint8)
            ebi, ebd, esd = oracle.bf_top2(descs[g], refs)
            nc = counts[g]
            assert np.array_equal(bi[off:off + nc], ebi) and np.array_equal(bd[off:off + nc], ebd) and np.array_equal(sd[off:off + nc], esd)
            off += nc
        assert off == len(bi)
    matcher.close()


@pytest.mark.parametrize("n_per_cam,nq,th,seed,occ_p", [([800, 600], 900, 10.0, 1, 0.1), ([1500, 1500], 3000, 20.0, 2, 0.0), ([300, 200], 1200, 10.0, 3, 0.3)])
def test_two_window_loop_search_equals_oracle(matcher, n_per_cam, nq, th, seed, occ_p, monkeypatch):
    """orbm_search_by_projection_windows = SearchByProjection(KeyFrame*, Scw, points, cams, vpMatched, th, Calib), reference
    src/ORBmatcher.cc:566-750: best candidate over the windows of BOTH cameras, camera 1's candidates first, accepted matches
    hide their feature from later points -- against the oracle's literal restatement, on the device resolve and on the
    exact host fallback."""
    import multi_orb_slam_amd as m
    fr = helpers.make_frame_arrays(n_per_cam, 640, 480, seed + 60, with_right=False)
    q, w2 = helpers.make_two_window_queries(fr, nq, seed + 70, th)
    occ = (helpers.rand_unit(sum(n_per_cam), seed + 80) < occ_p).astype(np.uint8) if occ_p else None
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    en, emo = oracle.search_by_projection_loop2(OF, q, w2, occ, 50)
    n, mo = matcher.SearchByProjectionWindows(F, q, w2, 50, occ)
    assert n == en and np.array_equal(mo, emo) and n > nq // 10
    cams_hit = np.asarray(fr["cam_of"])[np.flatnonzero(emo >= 0)]
    assert (cams_hit == 0).sum() > 20 and (cams_hit == 1).sum() > 20          # winners come from both cameras
    # a point whose camera-1 window is missing still matches through its second window, and the other way round
    only2 = np.flatnonzero((q["cam"] < 0) & (w2["cam"] >= 0)); only1 = np.flatnonzero((q["cam"] >= 0) & (w2["cam"] < 0))
    assert np.isin(emo[emo >= 0], only2).any() and np.isin(emo[emo >= 0], only1).any()
    F.close()
    monkeypatch.setenv("MORB_HOST_RESOLVE", "1")                                 # the host replay of the same loop
    mt2 = m.Matcher(0.8, True)
    F2 = mt2.frame(m.FrameData(**fr))
    n2, mo2 = mt2.SearchByProjectionWindows(F2, q, w2, 50, occ)
    assert n2 == en and np.array_equal(mo2, emo)
    F2.close(); mt2.close()


@pytest.mark.parametrize("counts", [[1000, 1000], [1000, 500], [64, 64], [63, 200, 1], [300, 0, 257, 129, 64, 511, 2, 190],
                                    [4000, 3900, 4031, 3968, 4000, 4027, 3999, 4001], [70], [5, 7]])
def test_cross_top2_both_forms_equal_brute_force(counts):
    """orbm_cross_top2 (reference analogue: the unrestricted inner loop of src/ORBmatcher.cc:287-321 between the cameras of a
    rig) in its matrix-core form and in its xor/popcount form: camera boundaries inside a 64-row tile and inside a 64-query
    wave, empty cameras, a single camera (nothing to match), duplicates across cameras (first index wins, the duplicate is the
    second best), the configs[4] size (8 x ~4000: several reference slices, whole tiles skipped as the wave's own camera)."""
    import multi_orb_slam_amd as m
    n = sum(counts)
    base = synth.descriptors(512, 77)
    descs = []
    for c, k in enumerate(counts):
        d = synth.perturbed_queries(base, seed=900 + c, flip_p=0.06)
        d = np.concatenate([d] * (k // len(d) + 1))[:k].copy() if k else np.zeros((0, 32), np.uint8)
        if k > len(base):
            d[len(base):] = synth.perturbed_queries(d[len(base):], seed=950 + c, flip_p=0.2)   # (no exact repeats inside a camera)
        descs.append(np.ascontiguousarray(d))
    if len(counts) >= 2 and counts[0] >= 40 and counts[1] >= 40:
        descs[1][7] = descs[0][31]; descs[1][33] = descs[0][31]          # exact duplicates across cameras
    fr = helpers.make_frame_arrays(counts, 640, 480, 5)
    fr["descs"] = descs
    exp = []
    for c in range(len(counts)):
        others = [descs[o] for o in range(len(counts)) if o != c]
        refs = np.concatenate(others) if others else np.zeros((0, 32), np.uint8)
        exp.append(oracle.bf_top2(descs[c], refs) if counts[c] else (np.zeros(0, np.int32),) * 3)
    ebi, ebd, esd = (np.concatenate([e[k] for e in exp]) for k in range(3))
    try:
        for on, fp4 in ((1, 1), (1, 0), (0, -1)):     # matrix cores with FP4 / int8 arithmetic (orbm_use_fp4_top2), xor + popcount
            m.Matcher.use_matrix_cores(on); m.Matcher.use_fp4_top2(fp4)
            mt = m.Matcher()
            F = mt.frame(m.FrameData(**fr))
            bi, bd, sd = mt.cross_top2(F)
            assert np.array_equal(bi, ebi) and np.array_equal(bd, ebd) and np.array_equal(sd, esd), "form %d / %d" % (on, fp4)
            F.close(); mt.close()
    finally:
        m.Matcher.use_matrix_cores(-1); m.Matcher.use_fp4_top2(-1)
    if len(counts) >= 2 and counts[0] >= 40 and counts[1] >= 40:
        assert ebd[31] == 0 and esd[31] == 0 and ebi[31] == 7           # camera-0 feature 31: first duplicate wins, second = 0
    assert n == len(ebi)


def test_gathered_buffer_with_a_corrupt_trailer_is_an_error_not_an_overrun():
    """A remote rank's count trailer is data from another process: counts that are negative or exceed the block's rows are
    clamped on the device (k_repack_gathered never leaves its block or the contiguous list) and the call reports ORB_E_ARG."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import rt
    from multi_orb_slam_amd.dist import pack_export_block
    cap_rows, world, cpr = 256, 3, 2
    block_bytes = cap_rows * 32 + 256
    descs = [synth.descriptors(40 + 10 * g, 300 + g) for g in range(world * cpr)]
    good = np.concatenate([pack_export_block([descs[r * cpr + c] for c in range(cpr)], cap_rows) for r in range(world)])
    matcher = m.Matcher()
    dev = rt.DeviceBuffer(good.nbytes)
    for bad_counts in ([250, 250], [-5, 40], [2 ** 30, 1]):
        buf = good.copy()
        t = 1 * block_bytes + cap_rows * 32                          # rank 1's trailer
        buf[t:t + 8] = np.array(bad_counts, np.int32).view(np.uint8)
        dev.upload(buf)
        with pytest.raises(m.OrbError) as e:
            matcher.cross_top2_gathered(dev.ptr, world, block_bytes, cap_rows, cpr, 0)
        assert e.value.code == -1 and "counts" in str(e.value)
    dev.upload(good)                                                 # the handle is still usable afterwards
    bi, bd, sd, cnt = matcher.cross_top2_gathered(dev.ptr, world, block_bytes, cap_rows, cpr, 2)
    assert cnt == [len(d) for d in descs] and len(bi) == len(descs[4]) + len(descs[5])
    ebi, ebd, esd = oracle.bf_top2(descs[4], np.concatenate([descs[o] for o in range(6) if o != 4]))
    assert np.array_equal(bi[:len(descs[4])], ebi) and np.array_equal(bd[:len(descs[4])], ebd)
    matcher.close()


@pytest.mark.parametrize("n_per_cam,nq,th,seed", [([1000, 1000], 2000, 7.5, 1), ([2000], 3000, 4.0, 2), ([300, 200, 250], 900, 10.0, 3)])
def test_project_best_equals_oracle(matcher, n_per_cam, nq, th, seed):
    """orbm_project_best: the independent nearest-candidate loop of SearchBySim3 (gate none) and Fuse (chi-square gate)."""
    import multi_orb_slam_amd as m
    fr = helpers.make_frame_arrays(n_per_cam, 640, 480, seed)
    q = helpers.make_queries(fr, nq, seed + 70, th=th)
    lvl = np.maximum(q["max_level"], 0)
    q["min_level"] = lvl - 1; q["max_level"] = lvl
    n = len(fr["un_x"])
    occ = (helpers.rand_unit(n, seed + 5) < 0.25).astype(np.uint8)
    sg = (1.0 / (np.float32(1.2) ** np.arange(8)) ** 2).astype(np.float32)
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    found = 0
    for gate in (0, 1, 2):
        for o in (None, occ):
            bi, bd = matcher.project_best(F, q, o, gate, sg)
            obi, obd = oracle.project_best(OF, q, o, gate, sg)
            assert np.array_equal(bi, obi) and np.array_equal(bd, obd), (gate, o is None)
            found += int((bi >= 0).sum())
    assert found > nq
    bi, bd = matcher.project_best(F, q[:0], None, 0, None)
    assert len(bi) == 0
    with pytest.raises(m.OrbError):
        matcher.project_best(F, q, None, 2, None)        # chi-square gate without its level table
    F.close()


@pytest.mark.parametrize("th_high,check_ori", [(100, True), (64, True), (50, False)])
def test_relocalisation_and_loop_style_searches(matcher, th_high, check_ori):
    """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (reference :3809-3946) and the loop overload
    (:753-867) on the tracking primitive: no right-coordinate gate (ur = NaN), every claim blocks, pre-assigned features
    hidden, ORBdist / TH_LOW as the acceptance threshold, histogram on or off."""
    import multi_orb_slam_amd as m
    fr = helpers.make_frame_arrays([1500], 640, 480, 11)
    q = helpers.make_queries(fr, 1800, 31, th=10.0, blocks=1)
    q["ur"] = np.nan; q["cam"] = 0
    lvl = np.maximum(q["max_level"], 0)
    q["min_level"] = lvl - 1; q["max_level"] = lvl + (1 if check_ori else 0)
    occ = (helpers.rand_unit(1500, 17) < 0.3).astype(np.uint8)
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    matcher.check_orientation = check_ori
    n, mo = matcher.SearchByProjection(F, q, th_high, occ)
    on, omo = oracle.search_by_projection_frames(OF, q, th_high, check_ori, occ)
    matcher.check_orientation = True
    assert n == on and np.array_equal(mo, omo) and n > 100
    assert not np.any((mo >= 0) & (occ != 0))
    F.close()


@pytest.mark.parametrize("nq,nr", [(64, 64), (65, 127), (256, 4033), (700, 3000), (1000, 191)])
def test_fp4_top2_edge_distances(matcher, nq, nr):
    """The FP4 form of the matrix-core top-2 turns a descriptor bit into +-4 (E2M1) and reads the f32 accumulator as the integer key
    32 * distance + row: the extremes of that arithmetic -- distance 0 (identical rows), 256 (complements), all-zero and all-one
    descriptors, ties between rows of one 32-row block and between blocks (the first index wins, the second best equals the best) --
    against the oracle's brute force, at sizes with ragged last tiles."""
    import multi_orb_slam_amd as m
    r = synth.descriptors(nr, 5)
    r[0] = 0; r[1] = 255; r[min(40, nr - 1)] = r[min(3, nr - 1)]; r[nr - 1] = r[2]      # duplicates inside a block and across blocks
    q = synth.perturbed_queries(np.concatenate([r] * (nq // nr + 1))[:nq], 9, flip_p=0.1)
    q[0] = 0; q[1] = 255; q[2] = ~r[5 % nr]; q[3] = r[2]; q[4] = r[3 % nr]
    ebi, ebd, esd = oracle.bf_top2(q, r)
    try:
        for fp4 in (1, 0):
            m.Matcher.use_fp4_top2(fp4)
            bi, bd, sd = matcher.hamming_top2(q, r)
            assert np.array_equal(bi, ebi) and np.array_equal(bd, ebd) and np.array_equal(sd, esd), fp4
    finally:
        m.Matcher.use_fp4_top2(-1)
    assert ebd[0] == 0 and ebd[1] == 0 and ebd[3] == 0 and ebi[3] == 2 and esd[3] == 0
