"""Known-answer tests for the ctor tables and the rBRIEF pattern.  CPU only."""
import zlib
import numpy as np
import oracle
import multi_orb_slam_amd as m


def test_pattern_table_pinned():
    p = oracle.pattern()
    assert p.shape == (256, 4)
    assert p[0].tolist() == [8, -3, 9, 5]        # reference src/ORBextractor.cc:152
    assert p[-1].tolist() == [-1, -6, 0, -11]    # reference src/ORBextractor.cc:407
    assert np.abs(p).max() == 13
    assert zlib.crc32(p.astype(np.int8).tobytes()) == 0xD1A39030
    # every rotated tap stays within +-19 px of the keypoint (EDGE_THRESHOLD): max radius
    rad = np.sqrt((p[:, 0::2].astype(float) ** 2 + p[:, 1::2].astype(float) ** 2)).max()
    assert rad < 18.5


def test_scale_tables_known_values():
    t = oracle.tables(1000, 1.2, 8, 20, 7)
    expect = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.allclose(t["scale"], np.array(expect, np.float32), rtol=0, atol=1e-7)
    assert np.array_equal(t["sigma2"], t["scale"] * t["scale"])
    assert np.array_equal(t["inv_scale"], np.float32(1) / t["scale"])
    assert np.array_equal(t["inv_sigma2"], np.float32(1) / t["sigma2"])


def test_feature_quotas_known_values():
    q = {1000: [217, 181, 151, 126, 105, 87, 73, 60], 500: [109, 90, 75, 63, 52, 44, 36, 31],
         2000: [434, 362, 302, 251, 209, 175, 145, 122], 4000: [869, 724, 603, 503, 419, 349, 291, 242]}
    for n, exp in q.items():
        got = oracle.tables(n, 1.2, 8, 20, 7)["quota"].tolist()
        assert got == exp and sum(got) == n


def test_umax_known_values():
    assert oracle.tables()["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # circular patch has 749 pixels
    u = oracle.tables()["umax"]
    assert 31 + 2 * sum(2 * int(u[v]) + 1 for v in range(1, 16)) == 749


def test_level_sizes_known_values():
    assert oracle.level_sizes(640, 480) == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193),
                                            (214, 161), (179, 134)]
    assert sum(w * h for w, h in oracle.level_sizes(640, 480)) == 950532
    assert sum(w * h for w, h in oracle.level_sizes(1280, 720)) == 2853088
    assert sum(w * h for w, h in oracle.level_sizes(1920, 1080)) == 6419321


def test_product_tables_match_oracle():
    """orbx_tables is host-only code in libmorb.so: must agree with the oracle for a sweep of ctor arguments."""
    for nf in (1, 50, 500, 1000, 1500, 2000, 4000):
        for sf in (1.2, 1.1, 1.5, 2.0):
            for nl in (1, 4, 8, 12):
                a = m.tables(m.ExtractorParams(nfeatures=nf, scale_factor=sf, nlevels=nl))
                b = oracle.tables(nf, sf, nl, 20, 7)
                for k in ("scale", "inv_scale", "sigma2", "inv_sigma2", "quota", "umax"):
                    assert np.array_equal(a[k], b[k]), (nf, sf, nl, k)
