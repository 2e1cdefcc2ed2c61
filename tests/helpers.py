"""Shared builders for matcher test inputs (used by CPU oracle tests and GPU parity tests)."""
import numpy as np
from multi_orb_slam_amd import synth
from multi_orb_slam_amd._lib import QUERY_DTYPE


def rand_u32(n, seed):
    return synth.hash32(np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B1) & 0xFFFFFFFF))


def rand_unit(n, seed):
    return rand_u32(n, seed).astype(np.float64) / 2.0 ** 32


def make_frame_arrays(n_per_cam, width, height, seed=1, nlevels=8, frac_coords=True, with_right=True):
    """Random 'current frame': keypoints spread over the image (some outside the bounds, some on cell edges)."""
    xs, ys, octs, angs, urs, cams, locs, descs = [], [], [], [], [], [], [], []
    for c, n in enumerate(n_per_cam):
        x = rand_unit(n, seed + 11 * c) * (width + 20) - 10
        y = rand_unit(n, seed + 11 * c + 1) * (height + 20) - 10
        if not frac_coords:
            x = np.floor(x); y = np.floor(y)
        o = (rand_u32(n, seed + 11 * c + 2) % nlevels).astype(np.int32)
        a = (rand_unit(n, seed + 11 * c + 3) * 360).astype(np.float32)
        ur = np.where(rand_unit(n, seed + 11 * c + 4) < 0.6, x - 5 - 40 * rand_unit(n, seed + 11 * c + 5), -1.0)
        if not with_right:
            ur = np.full(n, -1.0)
        xs.append(x.astype(np.float32)); ys.append(y.astype(np.float32)); octs.append(o); angs.append(a)
        urs.append(ur.astype(np.float32)); cams.append(np.full(n, c, np.int32)); locs.append(np.arange(n, dtype=np.int32))
        descs.append(synth.descriptors(n, seed + 100 + c))
    cat = np.concatenate
    return dict(un_x=cat(xs), un_y=cat(ys), octave=cat(octs), angle=cat(angs), uright=cat(urs), cam_of=cat(cams),
                local_of=cat(locs), descs=descs, bounds=(0.0, 0.0, float(width), float(height)))


def make_queries(fr, nq, seed=5, th=15.0, scale_factor=1.2, nlevels=8, dup_prob=0.5, blocks=1):
    """Projected 'last frame' points: half are perturbed copies of frame features (so real matches exist)."""
    n = len(fr["un_x"])
    q = np.zeros(nq, QUERY_DTYPE)
    pick = (rand_u32(nq, seed) % max(n, 1)).astype(np.int64)
    scales = (np.float32(scale_factor) ** np.arange(nlevels)).astype(np.float32)
    jx = (rand_unit(nq, seed + 1) - 0.5) * 12
    jy = (rand_unit(nq, seed + 2) - 0.5) * 12
    all_desc = np.concatenate(fr["descs"]) if n else np.zeros((1, 32), np.uint8)
    base_desc = all_desc[np.minimum(pick, len(all_desc) - 1)] if n else np.zeros((nq, 32), np.uint8)
    pert = synth.perturbed_queries(base_desc, seed + 3, 0.06)
    use_dup = rand_unit(nq, seed + 4) < dup_prob
    for i in range(nq):
        g = pick[i] if n else 0
        octv = int(fr["octave"][g]) if n else 0
        q["u"][i] = (fr["un_x"][g] if n else 100) + jx[i]
        q["v"][i] = (fr["un_y"][g] if n else 100) + jy[i]
        q["radius"][i] = np.float32(th) * scales[octv]
        q["ur"][i] = q["u"][i] - 20.0
        mode = i % 3
        q["min_level"][i], q["max_level"][i] = [(octv, -1), (0, octv), (octv - 1, octv + 1)][mode]
        q["cam"][i] = int(fr["cam_of"][g]) if n else 0
        q["blocks"][i] = blocks if blocks in (0, 1) else int(rand_u32(1, seed + i)[0] & 1)
        q["angle"][i] = (fr["angle"][g] if n else 0) + (rand_unit(1, seed + 50 + i)[0] - 0.5) * 80
        q["desc"][i] = base_desc[i] if use_dup[i] else pert[i]
    q["angle"] = np.mod(q["angle"], 360).astype(np.float32)
    return q


def make_bow_pair(voc, ovoc, n_a, n_b, seed=1, levelsup=2, n_cams=2, nlevels=8, flag_p=0.8, stereo_p=0.5):
    """Two synthetic keyframes for the BoW searches: B holds perturbed copies of many of A's descriptors (so that node-local
    nearest neighbours exist, some of them contested), angles of true pairs differ by a common rotation + jitter.
    ovoc: an object with bow_vectors(features, levelsup) -> (_, (node_id, node_start, items)).  -> (side_a, side_b) dicts."""
    from multi_orb_slam_amd import synth
    da = synth.vocabulary_words(voc, n_a, seed)
    src = rand_u32(n_b, seed + 1) % np.uint32(max(n_a, 1))
    db = synth._flip_bits(da[src], seed + 2, 0.04) if n_a else synth.vocabulary_words(voc, n_b, seed + 9)
    fresh = rand_unit(n_b, seed + 3) < 0.25
    db[fresh] = synth.vocabulary_words(voc, int(fresh.sum()), seed + 4)
    ang_a = (rand_unit(n_a, seed + 5) * 360.0).astype(np.float32)
    jitter = (rand_unit(n_b, seed + 6) - 0.5) * 20.0
    wild = rand_unit(n_b, seed + 7) < 0.15
    ang_b = np.mod(ang_a[src].astype(np.float64) - 40.0 + jitter + wild * rand_unit(n_b, seed + 8) * 360.0, 360.0).astype(np.float32)

    def side(desc, ang, n, s):
        (_, (nid, nstart, items)) = ovoc.bow_vectors(desc, levelsup)
        fl = (rand_unit(n, s) < flag_p).astype(np.uint8) | ((rand_unit(n, s + 1) < stereo_p).astype(np.uint8) << 1)
        return dict(desc=desc, angle=ang, flags=fl, node_id=nid, node_start=nstart, items=items,
                    x=(rand_unit(n, s + 2) * 640).astype(np.float32), y=(rand_unit(n, s + 3) * 480).astype(np.float32),
                    octave=(rand_u32(n, s + 4) % np.uint32(nlevels)).astype(np.int32),
                    cam_of=(rand_u32(n, s + 5) % np.uint32(n_cams)).astype(np.int32))
    sa, sb = side(da, ang_a, n_a, seed + 20), side(db, ang_b, n_b, seed + 40)
    if n_a:   # true pairs lie on (nearly) the same image row, 25 px apart: consistent with a sideways-translation F12
        sb["x"] = (sa["x"][src] + 25.0).astype(np.float32)
        sb["y"] = (sa["y"][src] + (rand_u32(n_b, seed + 60) % np.uint32(7)).astype(np.float32) - 3.0).astype(np.float32)
    return sa, sb


def make_two_window_queries(fr, nq, seed=5, th=10.0, scale_factor=1.2, nlevels=8):
    """Loop points of the two-camera loop search: every point looks at a feature of camera `c0` and (most of them) also at a
    feature of the other camera -- either of the two windows may be missing -- with a descriptor close to one of the two
    targets, so that the winner comes from either camera and contested features exist.  -> (queries, second windows)."""
    from multi_orb_slam_amd._lib import WINDOW_DTYPE
    n = len(fr["un_x"]); cam_of = np.asarray(fr["cam_of"])
    idx_by_cam = [np.flatnonzero(cam_of == c) for c in (0, 1)]
    q = np.zeros(nq, QUERY_DTYPE); w2 = np.zeros(nq, WINDOW_DTYPE)
    scales = (np.float32(scale_factor) ** np.arange(nlevels)).astype(np.float32)
    all_desc = np.concatenate(fr["descs"])
    r0 = rand_u32(nq, seed); r1 = rand_u32(nq, seed + 1); mode = rand_u32(nq, seed + 2) % 10
    jit = (rand_unit(4 * nq, seed + 3) - 0.5) * 8
    tgt = np.zeros(nq, np.int64)
    for i in range(nq):
        g0 = int(idx_by_cam[0][r0[i] % len(idx_by_cam[0])]); g1 = int(idx_by_cam[1][r1[i] % len(idx_by_cam[1])])
        lvl = int(fr["octave"][g0]) + int(mode[i] % 2)                  # predicted level: the feature's own or one above
        lvl = min(max(lvl, 0), nlevels - 1)
        rad = np.float32(th) * scales[lvl]
        q["u"][i] = fr["un_x"][g0] + jit[4 * i]; q["v"][i] = fr["un_y"][g0] + jit[4 * i + 1]; q["radius"][i] = rad
        q["min_level"][i] = lvl - 1; q["max_level"][i] = lvl; q["cam"][i] = 0 if mode[i] != 0 else -1     # 10 %: not visible in camera 1
        w2["u"][i] = fr["un_x"][g1] + jit[4 * i + 2]; w2["v"][i] = fr["un_y"][g1] + jit[4 * i + 3]; w2["radius"][i] = rad
        w2["min_level"][i] = lvl - 1; w2["max_level"][i] = lvl; w2["cam"][i] = 1 if mode[i] not in (1, 2) else -1   # 20 %: not in camera 2
        tgt[i] = g0 if (mode[i] % 3 and q["cam"][i] >= 0) else g1
    q["ur"] = np.nan; q["blocks"] = 1; q["angle"] = 0
    q["desc"] = synth.perturbed_queries(all_desc[tgt], seed + 4, 0.04)
    q["desc"][::3] = all_desc[tgt][::3]
    return q, w2
