#!/usr/bin/env python3
"""Turn rocprofv3 --pmc CSVs (FETCH_SIZE and WRITE_SIZE passes of tools/profile_matcher.py) into
profiles/pmc_traffic.json: HBM bytes per launch for each kernel, corrected as MI355X_MICROARCH.md section HBM prescribes
(FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2; both counters are in KiB), with the two calibration kernels
of known byte counts (1 GiB memset, 1 GiB device copy) reported next to them."""
import collections, csv, json, sys

def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for key in ("k_hamming_matrix_mfma", "k_hamming_top2_mfma", "k_cross_top2_mfma", "k_cross_top2", "k_hamming_matrix", "k_hamming_top2", "k_top2_merge", "k_project",
                    "k_resolve", "k_bow_transform", "k_bow_join", "fillBuffer", "copyBuffer"):
            if key in name:
                agg[(key, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
                break
    return {k: sum(v) / len(v) for k, v in agg.items()}

def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv), load(write_csv)
    GIB = float(1 << 30)
    cal_w = w[[k for k in w if k[0] == "fillBuffer"][0]] * 1024 / GIB
    cal_f = f[[k for k in f if k[0] == "copyBuffer"][0]] * 1024 / GIB
    out = {"_calibration": {"memset_1GiB_WRITE_SIZE_ratio": round(cal_w, 4), "copy_1GiB_FETCH_SIZE_ratio": round(cal_f, 4),
                            "note": "WRITE_SIZE reads 1.0 x the known bytes; FETCH_SIZE reads 0.5 x (gfx950) -> corrected x2"}}
    for k in sorted(set(f) | set(w)):
        fb = f.get(k, 0.0) * 1024 * 2.0
        wb = w.get(k, 0.0) * 1024
        name = k[0] if not k[0].startswith("k_hamming_top2") else "%s[grid=%d]" % k
        out[name] = {"grid_size": k[1], "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                     "hbm_bytes_per_launch": round(fb + wb)}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:4])
