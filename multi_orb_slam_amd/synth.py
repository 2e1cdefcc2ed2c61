"""Deterministic synthetic camera streams and matcher inputs (no images ship with the reference).

Counter-based integer hashing only, so the same bytes can be regenerated anywhere (numpy, C++) without an RNG
library: scene = mid-grey canvas + axis-aligned rectangles of random size/grey level (FAST corners at many scales),
frame t = the scene translated by (3t, t) px + fresh per-pixel noise in [-6, 6].
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def hash32(x):
    """murmur3 finaliser on uint32 arrays."""
    x = np.asarray(x, np.uint64) & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def _stream(seed, n):
    return hash32(np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF))


def scene_rects(cam, width, height):
    n = max(8, int(400 * (width * height) / 307200.0))
    r = _stream(0x51ED270B ^ (cam * 7919), 5 * n).astype(np.int64)
    margin = 96
    x0 = r[0::5] % (width + 2 * margin) - margin
    y0 = r[1::5] % (height + 2 * margin) - margin
    w = 4 + r[2::5] % 61
    h = 4 + r[3::5] % 61
    g = r[4::5] % 256
    return np.stack([x0, y0, w, h, g], 1)


def image(cam, t, width, height):
    """uint8 HxW frame t of camera `cam`."""
    img = np.full((height, width), 128, np.int32)
    dx, dy = 3 * t, t
    for x0, y0, w, h, g in scene_rects(cam, width, height):
        xa, ya = max(0, x0 + dx), max(0, y0 + dy)
        xb, yb = min(width, x0 + dx + w), min(height, y0 + dy + h)
        if xa < xb and ya < yb:
            img[ya:yb, xa:xb] = g
    idx = np.arange(width * height, dtype=np.uint64) + np.uint64(((cam * 100003 + t) * 2654435761) & 0xFFFFFFFF)
    noise = (hash32(idx) % np.uint32(13)).astype(np.int32).reshape(height, width) - 6
    return np.clip(img + noise, 0, 255).astype(np.uint8)


# ---- further image families (round 5): what the rectangle scenes never put in front of the kernels -- smooth gradients, soft
# edges, saturated regions and texture just under and just over the FAST thresholds, where the fixed-point roundings of the
# resize (>>4 .. >>16 .. +2 >>2) and of the blur ((s + 32768) >> 16) decide bytes.  Integer arithmetic only, like image().
FAMILIES = ("pink", "ramp", "soft", "saturated", "threshold")


def _box_blur(a, r):
    """(2r+1)^2 box mean of an int array, edge-replicated, rounded to nearest (integer arithmetic)."""
    p = np.pad(a.astype(np.int64), r, mode="edge")
    c = np.cumsum(np.cumsum(np.pad(p, ((1, 0), (1, 0))), axis=0), axis=1)
    k = 2 * r + 1
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return (s + k * k // 2) // (k * k)


def _white(seed, h, w, mod):
    return (hash32(np.arange(w * h, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFF)) % np.uint32(mod)).astype(np.int64).reshape(h, w)


def family_image(kind, cam, t, width, height):
    """uint8 HxW frame t of camera `cam` of one of FAMILIES:
    pink       1/f-like noise: white noise octaves, each box-blurred at twice the radius and weighted by it
    ramp       a smooth two-way gradient plus two broad quadratic bumps, dithered by +-1, the rectangle scene on top at a third of its contrast
    soft       the rectangle scene of image() behind a 5x5 or 9x9 box blur (soft edges), little noise
    saturated  the rectangle scene stretched so that a third of it clips at 0 and 255, with noise on top (clipped again)
    threshold  weak rectangles (contrast 5..26 around mid-grey: both sides of iniThFAST = 20 / minThFAST = 7) on +-3 texture"""
    seed = (cam * 100003 + t) * 2654435761 + 977 * FAMILIES.index(kind)
    if kind == "pink":
        acc = np.zeros((height, width), np.int64)
        wsum = 0
        for o, r in enumerate((0, 1, 2, 4, 8, 16)):
            n = _white(seed + 7919 * o, height, width, 256) - 128
            acc += (_box_blur(n, r) if r else n) * (2 * r + 1)
            wsum += 1
        img = 128 + acc // (2 * wsum)
    elif kind == "ramp":
        y, x = np.mgrid[0:height, 0:width].astype(np.int64)
        cx, cy = width // 3 + 5 * t, height // 2 + 3 * t
        bump = 90 - ((x - cx) ** 2 + (y - cy) ** 2) * 90 // (width * width // 9)
        bump2 = 70 - ((x - 2 * cx) ** 2 + (y - cy // 2) ** 2) * 70 // (width * width // 16)
        rects = (image(cam, t, width, height).astype(np.int64) - 128) // 3     # (something for FAST to find on the slopes)
        img = 20 + (x * 150) // width + (y * 60) // height + np.maximum(bump, 0) + np.maximum(bump2, 0) + rects + _white(seed, height, width, 3) - 1
    elif kind == "soft":
        base = image(cam, t, width, height).astype(np.int64)
        img = _box_blur(base, 2 if cam % 2 == 0 else 4) + _white(seed, height, width, 5) - 2
    elif kind == "saturated":
        base = image(cam, t, width, height).astype(np.int64)
        img = (base - 128) * 3 + 128 + _white(seed, height, width, 41) - 20
    elif kind == "threshold":
        img = np.full((height, width), 128, np.int64)
        dx, dy = 3 * t, t
        for x0, y0, w, h, g in scene_rects(cam, width, height):
            xa, ya = max(0, x0 + dx), max(0, y0 + dy)
            xb, yb = min(width, x0 + dx + w), min(height, y0 + dy + h)
            if xa < xb and ya < yb:
                img[ya:yb, xa:xb] = 128 + (5 + g % 22) * (1 if g & 64 else -1)
        img = img + _white(seed, height, width, 7) - 3
    else:
        raise ValueError(kind)
    return np.clip(img, 0, 255).astype(np.uint8)


def descriptors(n, seed=42):
    return hash32(np.arange(n * 8, dtype=np.uint64) + np.uint64((seed * 0x9E3779B1) & 0xFFFFFFFF)).view(np.uint8).reshape(n, 32).copy()


def perturbed_queries(refs, seed=7, flip_p=0.08):
    """Half of the rows = a reference row with each bit flipped with probability flip_p, the rest fresh random."""
    n = len(refs)
    bits = hash32(np.arange(n * 256, dtype=np.uint64) + np.uint64(seed * 77777)).astype(np.float64) / 2.0 ** 32 < flip_p
    flips = np.packbits(bits.reshape(n, 256), axis=1, bitorder="little")
    q = refs ^ flips
    fresh = descriptors(n, seed + 1000)
    half = (np.arange(n) % 2) == 1
    q[half] = fresh[half]
    return q


def _flip_bits(rows, seed, p):
    n = len(rows)
    bits = hash32(np.arange(n * 256, dtype=np.uint64) + np.uint64((seed * 0x2545F491) & 0xFFFFFFFF)).astype(np.float64) / 2.0 ** 32 < p
    return rows ^ np.packbits(bits.reshape(n, 256), axis=1, bitorder="little")


def _flip_eighth(rows, seed):
    """each bit flipped with probability 1/8 (three hashed words ANDed): cheap enough for a million rows"""
    n = len(rows)
    idx = np.arange(n * 8, dtype=np.uint64)
    m = hash32(idx + np.uint64((seed * 0x2545F491) & 0xFFFFFFFF)) & hash32(idx + np.uint64((seed * 0x9E3779B1 + 1) & 0xFFFFFFFF)) & \
        hash32(idx + np.uint64((seed * 0x85EBCA6B + 2) & 0xFFFFFFFF))
    return rows ^ m.view(np.uint8).reshape(n, 32)


def vocabulary(k=10, L=3, seed=11, ragged=False, stop_every=0):
    """Synthetic vocabulary tree in the layout of DBoW2's text loader (no vocabulary file ships with the reference): node ids in
    creation order, the children of a node = its descriptor with ~12 % of the bits flipped, idf-like positive weights.
    ragged: every 7th inner node becomes a leaf early and child counts vary in [2, k]; stop_every: every n-th word gets weight 0.
    -> dict(parent, is_leaf, desc, weight: arrays over node ids; k, L)."""
    if not ragged:   # full k-ary tree, level by level (vectorised: the stock k=10, L=6 shape has 1.1 M nodes)
        parent, leaf, desc, weight = [np.zeros(1, np.int64)], [np.zeros(1, np.uint8)], [np.zeros((1, 32), np.uint8)], [np.zeros(1)]
        ids = np.zeros(1, np.int64); dl = descriptors(1, seed); total = 1
        for lvl in range(L):
            cnt = len(ids) * k
            base = np.repeat(dl, k, 0) if lvl else descriptors(k, seed + 1)
            kids = _flip_eighth(base, seed + 13 * lvl + 1) if lvl else base
            cid = total + np.arange(cnt, dtype=np.int64)
            is_leaf = lvl + 1 == L
            parent.append(np.repeat(ids, k)); leaf.append(np.full(cnt, int(is_leaf), np.uint8)); desc.append(kids)
            wv = 0.5 + (hash32(cid.astype(np.uint64) + np.uint64(977 * seed)) % np.uint32(100000)).astype(np.float64) / 12345.0
            weight.append(wv if is_leaf else np.zeros(cnt))
            ids, dl, total = cid, kids, total + cnt
        parent, leaf, desc, weight = np.concatenate(parent), np.concatenate(leaf), np.concatenate(desc), np.concatenate(weight)
    else:
        parent, leaf, desc, weight = [0], [0], [np.zeros(32, np.uint8)], [0.0]
        root = descriptors(1, seed)[0]
        frontier = [(0, root, 0)]
        while frontier:
            nxt = []
            for (nid, d, lvl) in frontier:
                h = int(hash32(np.array([nid * 31 + seed], np.uint64))[0])
                kk = k if not ragged else 2 + h % (k - 1)
                base = np.repeat(d[None, :], kk, 0) if nid else descriptors(kk, seed + 1)
                kids = _flip_bits(base, seed + 13 * nid + 1, 0.12) if nid else base
                for c in range(kk):
                    cid = len(parent)
                    is_leaf = (lvl + 1 == L) or (ragged and cid % 7 == 3)
                    parent.append(nid); leaf.append(int(is_leaf)); desc.append(kids[c])
                    wv = 0.5 + (int(hash32(np.array([cid + 977 * seed], np.uint64))[0]) % 100000) / 12345.0
                    weight.append(wv if is_leaf else 0.0)
                    if not is_leaf:
                        nxt.append((cid, kids[c], lvl + 1))
            frontier = nxt
    weight = np.array(weight, np.float64); leaf = np.array(leaf, np.uint8)
    if stop_every:
        words = np.flatnonzero(leaf)
        weight[words[::stop_every]] = 0.0
    return dict(parent=np.array(parent, np.int32), is_leaf=leaf, desc=np.stack(desc).astype(np.uint8), weight=weight, k=k, L=L)


def vocabulary_words(voc, n, seed=3, flip_p=0.05):
    """n feature descriptors near random words of the vocabulary (so that descents reach varied leaves)."""
    words = np.flatnonzero(voc["is_leaf"])
    pick = words[hash32(np.arange(n, dtype=np.uint64) + np.uint64(seed * 1013)) % np.uint32(len(words))]
    return _flip_bits(voc["desc"][pick].copy(), seed + 5, flip_p)


def write_vocabulary_text(voc, path, trailing_blank=True):
    """The text format TemplatedVocabulary::saveToTextFile writes / loadFromTextFile reads."""
    with open(path, "w") as f:
        f.write("%d %d  0 0\n" % (voc["k"], voc["L"]))
        for i in range(1, len(voc["parent"])):
            f.write("%d %d %s %r\n" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]), float(voc["weight"][i])))
        if trailing_blank:
            f.write("\n")
