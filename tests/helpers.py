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
