#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs of tools/profile_extractor.py -> the "extractor" section of profiles/pmc_traffic.json.

usage: parse_pmc_extractor.py <dir with cfg{1,2,4}_{fetch,write,sq}/..._counter_collection.csv> <steps> <pmc_traffic.json>

Per configuration and kernel: launches per timestep, HBM bytes fetched / written per timestep (FETCH_SIZE and WRITE_SIZE are
in KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2, checked against the 1 GiB copy of the same run), and the SQ
counters per timestep.  `extract_chain` = the extraction kernels (k_ingest, k_resize2 / k_resize, k_fast_cells, k_octree,
k_describe; round 3: k_pyramid_tiled in place of the first three) summed, per IMAGE, next to SURVEY section 8(d)'s algorithmic bytes per image."""
import collections, csv, glob, json, os, sys

KERNELS = ["k_pyramid_tiled4", "k_pyramid_tiled", "k_ingest", "k_set_l0", "k_resize_v4", "k_resize2", "k_resize", "k_fast_cells", "k_octree", "k_describe", "k_frame_build_small", "k_frame_fill",
           "k_scan_cells", "k_scatter_cells", "k_sort_cells", "k_cams_from_counts", "k_project_side", "k_project", "k_top2_merge",
           "k_cross_top2_mfma", "k_resolve_mono", "k_resolve_cams", "k_resolve", "k_peer_put", "k_peer_wait", "k_rs_init", "k_rs_sweep", "k_rs_owner", "k_rs_reject", "k_rs_write", "k_mirror_frame",
           "fillBuffer", "copyBuffer"]
EXTRACT = ["k_pyramid_tiled4", "k_pyramid_tiled", "k_ingest", "k_set_l0", "k_resize_v4", "k_resize2", "k_resize", "k_fast_cells", "k_octree", "k_describe"]
SHAPES = {1: (640, 480, 1000, 2), 2: (1280, 720, 2000, 2), 3: (640, 480, 1000, 4), 4: (1920, 1080, 4000, 8)}


def short(name):
    for k in KERNELS:      # (k_resize_v4 / k_resize2 before k_resize, k_project_side before k_project, k_resolve_mono before k_resolve: first match wins)
        if k in name:
            return k
    return None


def load(path):
    """{kernel: {counter: [sum, launches]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            e = out[k][r["Counter_Name"]]
            e[0] += float(r["Counter_Value"]); e[1] += 1
    return out


def find(d, tag):
    g = glob.glob(os.path.join(d, tag, "**", "*counter_collection.csv"), recursive=True)
    return g[0] if g else None


def alg_bytes(w, h, nf, nlevels=8, scale=1.2):
    import numpy as np
    b = float(w * h); s = np.float32(1.0)
    for _ in range(1, nlevels):
        s = np.float32(s * np.float32(scale)); inv = np.float32(1.0) / s
        b += 2.0 * float(int(np.rint(np.float32(w) * inv))) * float(int(np.rint(np.float32(h) * inv)))
    return b + 60.0 * nf


def main(d, steps, out_json):
    steps = int(steps)
    doc = json.load(open(out_json)) if os.path.exists(out_json) else {}
    sec = {}
    for cfg, (w, h, nf, nc) in SHAPES.items():
        f, wr, sq = (find(d, "cfg%d_%s" % (cfg, t)) for t in ("fetch", "write", "sq"))
        if not f or not wr:
            continue
        F, Wt = load(f), load(wr)
        S = load(sq) if sq else {}
        cal_f = F["copyBuffer"]["FETCH_SIZE"][0] * 1024 / float(1 << 30) if "copyBuffer" in F else None
        cal_w = Wt["fillBuffer"]["WRITE_SIZE"][0] * 1024 / float(1 << 30) if "fillBuffer" in Wt else None
        fcorr = 2.0 if cal_f is None or cal_f < 0.75 else 1.0     # gfx950: FETCH_SIZE reads 0.5 x the known bytes
        entry = {"workload": "%d x %dx%d @%d, %d isolated timesteps" % (nc, w, h, nf, steps),
                 "_calibration": {"copy_1GiB_FETCH_SIZE_ratio": None if cal_f is None else round(cal_f, 4),
                                  "memset_1GiB_WRITE_SIZE_ratio": None if cal_w is None else round(cal_w, 4), "fetch_correction": fcorr},
                 "kernels": {}}
        chain_f = chain_w = 0.0
        for k in KERNELS:
            if k in ("fillBuffer", "copyBuffer") or (k not in F and k not in Wt):
                continue
            fb = F.get(k, {}).get("FETCH_SIZE", [0.0, 0])
            wb = Wt.get(k, {}).get("WRITE_SIZE", [0.0, 0])
            e = {"launches_per_step": round(max(fb[1], wb[1]) / steps, 2),
                 "fetch_bytes_per_step": round(fb[0] * 1024 * fcorr / steps), "write_bytes_per_step": round(wb[0] * 1024 / steps)}
            e["hbm_bytes_per_step"] = e["fetch_bytes_per_step"] + e["write_bytes_per_step"]
            if k in S:
                e["sq_per_step"] = {c: round(v[0] / steps) for c, v in sorted(S[k].items())}
            entry["kernels"][k] = e
            if k in EXTRACT:
                chain_f += e["fetch_bytes_per_step"]; chain_w += e["write_bytes_per_step"]
        alg = alg_bytes(w, h, nf)
        entry["extract_chain"] = {"kernels": EXTRACT, "fetch_bytes_per_image": round(chain_f / nc), "write_bytes_per_image": round(chain_w / nc),
                                  "hbm_bytes_per_image": round((chain_f + chain_w) / nc), "algorithmic_bytes_per_image": round(alg),
                                  "traffic_over_algorithmic": round((chain_f + chain_w) / nc / alg, 3)}
        sec["configs[%d]" % cfg] = entry
    doc["extractor"] = sec
    if "configs[1]" in sec:   # what bench.py's roofline_extract quotes as `traffic` (per image of the default configuration)
        doc["extract_chain"] = {"hbm_bytes_per_launch": sec["configs[1]"]["extract_chain"]["hbm_bytes_per_image"],
                                "unit": "bytes per 640x480 image (configs[1]), all extraction kernels"}
    json.dump(doc, open(out_json, "w"), indent=1)
    print(json.dumps(sec, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
