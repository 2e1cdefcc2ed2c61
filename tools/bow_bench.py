#!/usr/bin/env python3
"""Timings of the BoW row on the GPU next to the oracle on the host: vocabulary transform (stock shape k=10, L=6, 1.1 M nodes),
SearchByBoW (both overloads) and SearchForTriangulation at the feature counts of BASELINE.json's configs.  Prints one JSON line
per measurement; every GPU result is first compared with the oracle."""
import json
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import multi_orb_slam_amd as m  # noqa: E402
import oracle  # noqa: E402
from helpers import make_bow_pair  # noqa: E402
from multi_orb_slam_amd import synth, rt  # noqa: E402


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    voc = synth.vocabulary(10, 6, seed=3)
    t0 = time.perf_counter(); V = m.Vocabulary(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], voc["L"]); t_load = time.perf_counter() - t0
    O = oracle.Vocabulary(voc)
    print(json.dumps({"vocabulary": V.info(), "upload_s": round(t_load, 3)}))
    S = m.BowSearch()
    for n in (2000, 4000, 8000, 32000):
        feats = synth.vocabulary_words(voc, n, seed=5)
        w, nd, wt = V.transform(feats, 4); ow, ond, owt = O.transform(feats, 4)
        assert np.array_equal(w, ow) and np.array_equal(nd, ond) and np.array_equal(wt, owt)
        # kernel alone on resident descriptors (HIP events on the vocabulary's stream)
        d_f = rt.DeviceBuffer(feats.nbytes); d_w = rt.DeviceBuffer(4 * n); d_n = rt.DeviceBuffer(4 * n); d_f.upload(feats)
        for _ in range(3):
            V.transform_device(d_f.ptr, n, 4, d_w.ptr, d_n.ptr, V.stream)
        rt.stream_sync(V.stream)
        e0, e1 = rt.Event(), rt.Event(); e0.record(V.stream)
        for _ in range(50):
            V.transform_device(d_f.ptr, n, 4, d_w.ptr, d_n.ptr, V.stream)
        e1.record(V.stream); k_ms = e0.elapsed_ms(e1) / 50
        host_ms = timeit(lambda: V.transform(feats, 4), 20)
        bow_ms = timeit(lambda: V.bow_vectors(feats, 4), 20)
        cpu_ms = timeit(lambda: O.transform(feats, 4), 2)
        cpu_bow_ms = timeit(lambda: O.bow_vectors(feats, 4), 2)
        print(json.dumps({"op": "transform", "features": n, "kernel_ms": round(k_ms, 4), "host_call_ms": round(host_ms, 3),
                          "bow_vectors_ms": round(bow_ms, 3), "oracle_transform_ms": round(cpu_ms, 2), "oracle_bow_vectors_ms": round(cpu_bow_ms, 2),
                          "hamming_per_s": round(n * 60 / (k_ms * 1e-3) / 1e9, 2)}))
        for b in (d_f, d_w, d_n):
            b.free()
        a, bb = make_bow_pair(voc, O, n, n, seed=7, levelsup=4)
        fv = lambda s: m.FeatureVector(s["node_id"], s["node_start"], s["items"])
        A = m.BowSide(a["desc"], a["angle"], fv(a), a["flags"], a["x"], a["y"], a["octave"], a["cam_of"])
        B = m.BowSide(bb["desc"], bb["angle"], fv(bb), bb["flags"], bb["x"], bb["y"], bb["octave"], bb["cam_of"])
        KA, KB = S.keyframe(A), S.keyframe(B)
        for mode in (0, 1):
            nm, mt = S.search_by_bow(A, B, mode); onm, omt = oracle.search_by_bow(a, bb, mode)
            assert nm == onm and np.array_equal(mt, omt)
            rn, rm = S.search_by_bow_resident(KA, KB, mode, a["flags"], bb["flags"])
            assert rn == onm and np.array_equal(rm, omt)
            r_ms = timeit(lambda: S.search_by_bow_resident(KA, KB, mode, a["flags"], bb["flags"]), 50)
            g_ms = timeit(lambda: S.search_by_bow(A, B, mode), 20)
            c_ms = timeit(lambda: oracle.search_by_bow(a, bb, mode), 3)
            print(json.dumps({"op": "search_by_bow", "mode": mode, "features": n, "nodes": len(a["node_id"]), "largest_node": int(np.diff(bb["node_start"]).max()),
                              "matches": nm, "gpu_call_ms": round(g_ms, 3), "gpu_resident_call_ms": round(r_ms, 3), "oracle_ms": round(c_ms, 2)}))
        sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32); s2 = (sf * sf).astype(np.float32)
        F12 = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [1e-5, 0, 0.004, 0, 2e-5, -1, -0.004, 1, 0.3]], np.float32)
        ex, ey = np.array([300.0, -50.0], np.float32), np.array([200.0, 240.0], np.float32)
        nm, mt = S.search_for_triangulation(A, B, F12, ex, ey, sf, s2); onm, omt = oracle.search_for_triangulation(a, bb, F12, ex, ey, sf, s2)
        assert nm == onm and np.array_equal(mt, omt)
        r_ms = timeit(lambda: S.search_for_triangulation_resident(KA, KB, F12, ex, ey, sf, s2, a["flags"], bb["flags"]), 50)
        g_ms = timeit(lambda: S.search_for_triangulation(A, B, F12, ex, ey, sf, s2), 20)
        c_ms = timeit(lambda: oracle.search_for_triangulation(a, bb, F12, ex, ey, sf, s2), 3)
        print(json.dumps({"op": "search_for_triangulation", "features": n, "matches": nm, "gpu_call_ms": round(g_ms, 3), "gpu_resident_call_ms": round(r_ms, 3), "oracle_ms": round(c_ms, 2)}))
        KA.close(); KB.close()


if __name__ == "__main__":
    main()
