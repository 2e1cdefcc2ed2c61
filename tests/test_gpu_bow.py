"""HIP vocabulary transform and BoW-gated searches (include/orbv.h) against the oracle, bit-exact, through the C ABI."""
import numpy as np
import pytest
import torch  # noqa: F401  (first: torch ships its own HIP runtime)
import multi_orb_slam_amd as m
import oracle
from helpers import make_bow_pair
from multi_orb_slam_amd import synth

pytestmark = pytest.mark.gpu


def product_vocab(voc):
    return m.Vocabulary(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], voc["L"])


@pytest.mark.parametrize("k,L,ragged", [(10, 3, False), (10, 4, False), (17, 2, False), (9, 4, True), (2, 6, False)])
def test_transform_equals_oracle(k, L, ragged):
    voc = synth.vocabulary(k, L, seed=k + L, ragged=ragged, stop_every=5)
    V, O = product_vocab(voc), oracle.Vocabulary(voc)
    info = V.info()
    assert info["n_nodes"] == len(voc["parent"]) and info["n_words"] == int(voc["is_leaf"].sum()) and info["L"] == L
    feats = np.concatenate([synth.vocabulary_words(voc, 1500, seed=3), synth.descriptors(501, 77), voc["desc"][1:40]])
    for levelsup in (0, 1, 2, 4, L, L + 2):
        w, nd, wt = V.transform(feats, levelsup)
        ow, ond, owt = O.transform(feats, levelsup)
        assert np.array_equal(w, ow) and np.array_equal(nd, ond) and np.array_equal(wt, owt), levelsup
    V.close()


def test_transform_ties_take_the_first_child():
    d = lambda v: np.full(32, v, np.uint8)
    voc = dict(parent=np.array([0, 0, 0, 1, 1, 2, 2], np.int32), is_leaf=np.array([0, 0, 0, 1, 1, 1, 1], np.uint8),
               desc=np.stack([d(0), d(0x00), d(0xFF), d(0x00), d(0x0F), d(0xFF), d(0xF0)]),
               weight=np.array([0, 0, 0, 1.5, 2.25, 0.0, 4.0]), k=2, L=2)
    V = product_vocab(voc)
    f = np.stack([d(0x01), d(0x0F), d(0xFE), d(0xF0)])
    w, nd, wt = V.transform(f, 1)
    assert w.tolist() == [0, 1, 2, 0] and nd.tolist() == [1, 1, 2, 1] and wt.tolist() == [1.5, 2.25, 0.0, 1.5]
    V.close()


def test_node_ids_need_not_be_grouped_by_parent(tmp_path):
    """The loader accepts any file order (children = ascending id): shuffle the node ids of a tree and compare."""
    voc = synth.vocabulary(6, 3, seed=21)
    n = len(voc["parent"])
    perm = np.concatenate([[0], 1 + np.argsort(synth.hash32(np.arange(n - 1, dtype=np.uint64) + np.uint64(99)))])   # new id -> old id
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    # a child may now precede its parent in id order; the loader's invariant is only that the parent exists
    shuf = dict(parent=inv[voc["parent"][perm]].astype(np.int32), is_leaf=voc["is_leaf"][perm], desc=voc["desc"][perm],
                weight=voc["weight"][perm], k=6, L=3)
    shuf["parent"][0] = 0
    V, O = product_vocab(shuf), oracle.Vocabulary(shuf)
    feats = synth.vocabulary_words(voc, 800, seed=4)
    for a, b in zip(V.transform(feats, 1), O.transform(feats, 1)):
        assert np.array_equal(a, b)
    V.close()


def test_text_loader_round_trip(tmp_path):
    voc = synth.vocabulary(7, 3, seed=2, ragged=True, stop_every=4)
    path = tmp_path / "voc.txt"
    synth.write_vocabulary_text(voc, path, trailing_blank=True)
    V, O = m.Vocabulary(path=path), oracle.Vocabulary(voc)
    assert V.info()["n_nodes"] == len(voc["parent"])
    feats = synth.vocabulary_words(voc, 700, seed=6)
    for a, b in zip(V.transform(feats, 2), O.transform(feats, 2)):
        assert np.array_equal(a, b)
    V.close()
    bad = tmp_path / "bad.txt"; bad.write_text("10 6 1 0\n")
    with pytest.raises(m.OrbError):
        m.Vocabulary(path=bad)


def test_bow_vectors_are_bit_identical():
    voc = synth.vocabulary(10, 3, seed=9, stop_every=7)
    V, O = product_vocab(voc), oracle.Vocabulary(voc)
    for n, seed in ((2000, 1), (37, 2), (1, 3), (0, 4)):
        feats = synth.vocabulary_words(voc, n, seed=seed, flip_p=0.02)
        (ids, vals), fv = V.bow_vectors(feats, 2)
        (oids, ovals), (onid, onstart, oitems) = O.bow_vectors(feats, 2)
        assert np.array_equal(ids, oids) and vals.tobytes() == ovals.tobytes()
        assert np.array_equal(fv.node_id, onid) and np.array_equal(fv.node_start, onstart) and np.array_equal(fv.items, oitems)
        if n > 1:
            assert len(ids) < n           # words are shared: the one-by-one accumulation is exercised
            assert m.score_l1((ids, vals), (ids, vals)) == oracle.bow_score_l1((oids, ovals), (oids, ovals))
    a = V.bow_vectors(synth.vocabulary_words(voc, 900, seed=11), 2)[0]
    b = V.bow_vectors(synth.vocabulary_words(voc, 900, seed=12), 2)[0]
    assert m.score_l1(a, b) == oracle.bow_score_l1(a, b) and 0.0 <= m.score_l1(a, b) < 1.0
    V.close()


def test_transform_device_on_resident_descriptors():
    from multi_orb_slam_amd import rt
    voc = synth.vocabulary(10, 3, seed=4)
    V, O = product_vocab(voc), oracle.Vocabulary(voc)
    feats = synth.vocabulary_words(voc, 3000, seed=8)
    d_f = rt.DeviceBuffer(feats.nbytes); d_w = rt.DeviceBuffer(4 * len(feats)); d_n = rt.DeviceBuffer(4 * len(feats))
    d_f.upload(feats)
    V.transform_device(d_f.ptr, len(feats), 2, d_w.ptr, d_n.ptr, V.stream)
    rt.stream_sync(V.stream)
    ow, ond, _ = O.transform(feats, 2)
    assert np.array_equal(d_w.download(np.uint32, len(feats), V.stream), ow)
    assert np.array_equal(d_n.download(np.uint32, len(feats), V.stream), ond)
    for b in (d_f, d_w, d_n):
        b.free()
    V.close()


def to_side(s, tri=False):
    fv = m.FeatureVector(s["node_id"], s["node_start"], s["items"])
    if tri:
        return m.BowSide(s["desc"], s["angle"], fv, s["flags"], s["x"], s["y"], s["octave"], s["cam_of"])
    return m.BowSide(s["desc"], s["angle"], fv, s["flags"])


@pytest.mark.parametrize("k,levelsup,na,nb", [(10, 2, 1500, 1600), (10, 3, 2000, 1900), (3, 2, 1200, 1300), (10, 2, 40, 3000)])
def test_search_by_bow_equals_oracle(k, levelsup, na, nb):
    """(3, 2): three nodes with ~400 features each -- several 64-candidate chunks per query and long claim chains."""
    voc = synth.vocabulary(k, 3, seed=k)
    O = oracle.Vocabulary(voc)
    S = m.BowSearch()
    a, b = make_bow_pair(voc, O, na, nb, seed=na % 97, levelsup=levelsup)
    A, B = to_side(a), to_side(b)
    total = 0
    for mode in (0, 1):
        for (th, ratio, ori) in ((50, 0.7, True), (50, 0.75, False), (50, 0.9, True), (30, 0.6, True), (256, 1.0, True)):
            nm, match = S.search_by_bow(A, B, mode, th, ratio, ori)
            onm, omatch = oracle.search_by_bow(a, b, mode, th, ratio, ori)
            assert nm == onm and np.array_equal(match, omatch), (mode, th, ratio, ori)
            total += nm
    assert total > 100
    S.close()


def test_search_by_bow_edges():
    voc = synth.vocabulary(5, 3, seed=3)
    O = oracle.Vocabulary(voc)
    S = m.BowSearch()
    a, b = make_bow_pair(voc, O, 300, 320, seed=5, levelsup=2)
    # no flags array at all = every feature usable
    a1 = dict(a, flags=np.ones(300, np.uint8)); b1 = dict(b, flags=np.ones(320, np.uint8))
    A = m.BowSide(a["desc"], a["angle"], m.FeatureVector(a["node_id"], a["node_start"], a["items"]))
    B = m.BowSide(b["desc"], b["angle"], m.FeatureVector(b["node_id"], b["node_start"], b["items"]))
    for mode in (0, 1):
        nm, match = S.search_by_bow(A, B, mode)
        onm, omatch = oracle.search_by_bow(a1, b1, mode)
        assert nm == onm and np.array_equal(match, omatch)
    # nothing usable on the query side / disjoint node sets / empty sides
    a0 = dict(a, flags=np.zeros(300, np.uint8))
    nm, match = S.search_by_bow(to_side(a0), to_side(b), 0)
    assert nm == 0 and (match == -1).all()
    bd = dict(b, node_id=(b["node_id"] + np.uint32(100000)).astype(np.uint32))
    nm, match = S.search_by_bow(to_side(a), to_side(bd), 1)
    assert nm == 0 and (match == -1).all()
    empty = m.BowSide(np.zeros((0, 32), np.uint8), np.zeros(0, np.float32), m.FeatureVector([], [0], []))
    nm, match = S.search_by_bow(empty, to_side(b), 0)
    assert nm == 0 and len(match) == 320 and (match == -1).all()
    nm, match = S.search_by_bow(to_side(a), empty, 0)
    assert nm == 0 and len(match) == 0
    # malformed feature vector: loud error
    bad = dict(b, node_id=b["node_id"][::-1].copy())
    with pytest.raises(m.OrbError):
        S.search_by_bow(to_side(a), to_side(bad), 0)
    S.close()


@pytest.mark.parametrize("k,na,nb,stereo_p", [(10, 1500, 1600, 0.3), (3, 900, 1000, 0.0), (10, 2000, 2100, 1.0)])
def test_search_for_triangulation_equals_oracle(k, na, nb, stereo_p):
    voc = synth.vocabulary(k, 3, seed=k + 1)
    O = oracle.Vocabulary(voc)
    S = m.BowSearch()
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32); s2 = (sf * sf).astype(np.float32)
    a, b = make_bow_pair(voc, O, na, nb, seed=7, levelsup=2, stereo_p=stereo_p)
    F12 = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [1e-5, 0, 0.004, 0, 2e-5, -1, -0.004, 1, 0.3]], np.float32)
    ex, ey = np.array([300.0, -50.0], np.float32), np.array([200.0, 240.0], np.float32)
    A, B = to_side(a, True), to_side(b, True)
    total = 0
    for th, ori in ((50, True), (50, False), (35, True)):
        nm, match = S.search_for_triangulation(A, B, F12, ex, ey, sf, s2, th, ori)
        onm, omatch = oracle.search_for_triangulation(a, b, F12, ex, ey, sf, s2, th, ori)
        assert nm == onm and np.array_equal(match, omatch), (th, ori)
        total += nm
    assert total > 50
    S.close()


def test_resident_keyframes_equal_oracle_with_flags_of_the_moment():
    """Keyframes uploaded once, searched repeatedly while the MapPoint flags change between calls."""
    voc = synth.vocabulary(10, 3, seed=6)
    O = oracle.Vocabulary(voc)
    S = m.BowSearch()
    a, b = make_bow_pair(voc, O, 1700, 1800, seed=13, levelsup=2, stereo_p=0.3)
    KA, KB = S.keyframe(to_side(a, True)), S.keyframe(to_side(b, True))
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32); s2 = (sf * sf).astype(np.float32)
    F12 = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [1e-5, 0, 0.004, 0, 2e-5, -1, -0.004, 1, 0.3]], np.float32)
    ex, ey = np.array([300.0, -50.0], np.float32), np.array([200.0, 240.0], np.float32)
    from helpers import rand_unit
    for it in range(3):
        if it == 0:
            fa = fb = None; ea, eb = a, b                      # the flags uploaded with the keyframes
        else:
            fa = ((rand_unit(1700, 50 + it) < 0.7).astype(np.uint8) | (a["flags"] & 2)).astype(np.uint8)
            fb = ((rand_unit(1800, 60 + it) < 0.7).astype(np.uint8) | (b["flags"] & 2)).astype(np.uint8)
            ea, eb = dict(a, flags=fa), dict(b, flags=fb)
        for mode in (0, 1):
            nm, match = S.search_by_bow_resident(KA, KB, mode, fa, fb, 50, 0.75, True)
            onm, omatch = oracle.search_by_bow(ea, eb, mode, 50, 0.75, True)
            assert nm == onm and np.array_equal(match, omatch) and nm > 50
        nm, match = S.search_for_triangulation_resident(KA, KB, F12, ex, ey, sf, s2, fa, fb)
        onm, omatch = oracle.search_for_triangulation(ea, eb, F12, ex, ey, sf, s2)
        assert nm == onm and np.array_equal(match, omatch) and nm > 20
    # a keyframe without the triangulation arrays cannot be triangulated against: loud error
    KC = S.keyframe(to_side(a))
    with pytest.raises(m.OrbError):
        S.search_for_triangulation_resident(KC, KB, F12, ex, ey, sf, s2)
    for k in (KA, KB, KC):
        k.close()
    S.close()


def test_keyframes_built_on_the_device_from_the_front_end():
    """extract -> frame -> descents -> FeatureVector -> BoW searches without the features ever leaving HBM
    (orbf_export_features + orbv_keyframe_from_device); every stage against the oracle on the downloaded copies."""
    from multi_orb_slam_amd import pipeline
    voc = synth.vocabulary(10, 3, seed=12, stop_every=6)
    V, O = product_vocab(voc), oracle.Vocabulary(voc)
    S = m.BowSearch()
    params = [m.ExtractorParams(nfeatures=600), m.ExtractorParams(nfeatures=400)]
    fe = pipeline.FrontEnd(params, 320, 240)
    sides, kfs = [], []
    for t in range(2):
        got = fe.step([synth.image(c, 3 * t, 320, 240) for c in range(2)])
        feats = fe.fe.export_features()
        assert feats.n_total == sum(got["counts"]) == len(got["desc"]) and feats.n_total > 500
        kf = S.keyframe_from_device(V, feats, levelsup=2)
        w, nd, fv = S.keyframe_download(kf)
        ow, ond, _ = O.transform(got["desc"], 2)
        (_, (onid, onstart, oitems)) = O.bow_vectors(got["desc"], 2)
        assert np.array_equal(w, ow) and np.array_equal(nd, ond)
        assert np.array_equal(fv.node_id, onid) and np.array_equal(fv.node_start, onstart) and np.array_equal(fv.items, oitems)
        assert len(oitems) < feats.n_total          # stopped words were dropped
        n = feats.n_total
        cam_of = np.repeat(np.arange(2), got["counts"]).astype(np.int32)
        sides.append(dict(desc=got["desc"], angle=got["kps"]["angle"], flags=(1 | ((got["uright"] >= 0) << 1)).astype(np.uint8), node_id=onid,
                          node_start=onstart, items=oitems, x=got["un_x"], y=got["un_y"], octave=got["kps"]["octave"], cam_of=cam_of))
        kfs.append(kf)
    a, b = sides
    for mode in (0, 1):
        nm, match = S.search_by_bow_resident(kfs[0], kfs[1], mode, None, None, 50, 0.8, True)
        onm, omatch = oracle.search_by_bow(a, b, mode, 50, 0.8, True)
        assert nm == onm and np.array_equal(match, omatch)
    assert nm > 30
    sf = oracle.tables()["scale"]; s2 = (sf * sf).astype(np.float32)
    F12 = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0]] * 2, np.float32)       # rows of the two frames line up (the scene moves 9 px in x, 3 in y)
    ex, ey = np.array([-500.0, -500.0], np.float32), np.array([120.0, 120.0], np.float32)
    fl_a = (a["flags"] & 2) | 1; fl_b = (b["flags"] & 2) | 1
    nm, match = S.search_for_triangulation_resident(kfs[0], kfs[1], F12, ex, ey, sf, s2, fl_a, fl_b)
    onm, omatch = oracle.search_for_triangulation(dict(a, flags=fl_a), dict(b, flags=fl_b), F12, ex, ey, sf, s2)
    assert nm == onm and np.array_equal(match, omatch)
    for k in kfs:
        k.close()
    fe.close(); S.close(); V.close()


def test_bow_vectors_match_reference_classes_golden():
    # tests/golden/dbow2_ref_vectors.npz: outputs of the reference's own compiled BowVector / FeatureVector classes
    import os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_dbow2_golden as gold
    G = np.load(os.path.join(here, "golden", "dbow2_ref_vectors.npz"))
    import multi_orb_slam_amd as m
    for i, c in enumerate(gold.CASES):
        voc, feats = gold.case_inputs(c)
        V = m.Vocabulary(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], voc["L"])
        word, node, weight = V.transform(feats, c["levelsup"])
        assert np.array_equal(word, G["c%d_word" % i]) and np.array_equal(node, G["c%d_node" % i]) and np.array_equal(weight, G["c%d_weight" % i])
        (bid, bval), fv = V.bow_vectors(feats, c["levelsup"])
        assert np.array_equal(bid, G["c%d_bow_id" % i]) and np.array_equal(bval, G["c%d_bow_val" % i])
        assert np.array_equal(fv.node_id, G["c%d_fv_node" % i]) and np.array_equal(fv.node_start, G["c%d_fv_start" % i])
        assert np.array_equal(fv.items, G["c%d_fv_items" % i])
