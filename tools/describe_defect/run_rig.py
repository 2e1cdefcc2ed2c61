#!/usr/bin/env python3
"""Reproducer of the rare wrong descriptor bit (DESIGN.md "Known defect"): the loopback rig of tests/test_gpu_frontend.py -- four ranks
as four host threads on ONE GPU, three extractor streams + a matcher stream each, a late peer -- run for a fixed time; every
descriptor is held against the oracle's.  Prints one JSON line: runs, bad runs, the wrong bits seen (lane = bit // 4, j = bit % 4) and,
when the library is a self-check build (csrc/Makefile VARIANT=slp_check / check), k_describe's own counters.

    MORB_LIB_PATH=multi_orb_slam_amd/lib/libmorb_slp.so python tools/describe_defect/run_rig.py --seconds 150 [--tag name]
"""
import argparse, ctypes, json, os, sys, threading, time
root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--max-runs", type=int, default=100000)
ap.add_argument("--tag", default="")
ap.add_argument("--world", type=int, default=4)
ap.add_argument("--placement", default="inline")
ap.add_argument("--out", default="")
ap.add_argument("--persistent", type=int, default=0, help="N > 0: every rank keeps ONE front end for N runs in a row (reset in between) -- no handle is created or destroyed while others extract")
ap.add_argument("--no-probe", action="store_true", help="no exchange timing probe")
ap.add_argument("--no-delay", action="store_true", help="no late rank")
ap.add_argument("--no-exchange", action="store_true", help="the ranks' front ends run side by side without any exchange")
ap.add_argument("--all-delay", action="store_true", help="every rank sleeps 1 ms before every step (not only rank 1)")
args = ap.parse_args()
os.environ["MORB_EXCHANGE_PLACEMENT"] = args.placement
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, synth, _lib
from multi_orb_slam_amd.dist import shard_cameras
import oracle

world, n_cams, w, h, nf, ahead, T = args.world, args.world, 640, 480, 1000, 2, 10
frames = [{g: synth.image(g, t, w, h) for g in range(n_cams)} for t in range(T)]
exp = [[oracle.extract(frames[t][shard_cameras(n_cams, world, r)[0]], nfeatures=nf) for t in range(T)] for r in range(world)]
lib = m.lib()
check = getattr(lib, "morb_debug_describe_check", None)

def read_check():
    buf = (ctypes.c_ulonglong * (8 + 16 * 14 + 20))()
    assert check(buf, len(buf)) == 0
    return list(buf)

runs = bad_runs = 0
bits_seen, bad_detail = [], []
steps_bad = []
t_start = time.time()
inner = max(args.persistent, 1)


def check_run(results, run):
    bad = False
    for r in range(world):
        for t in range(T):
            got = results[r][t]
            ek, ed = exp[r][t]
            if len(ed) != len(got["desc"]) or got["kps"].tobytes() != ek.tobytes():
                bad = True; bad_detail.append({"run": run, "rank": r, "step": t, "what": "keypoints"}); continue
            diff = got["desc"] != ed
            if diff.any():
                bad = True; steps_bad.append(t)
                for i in np.flatnonzero(diff.any(axis=1)):
                    b = np.flatnonzero(np.unpackbits(got["desc"][i] ^ ed[i], bitorder="little")).tolist()
                    bits_seen.append(b)
                    bad_detail.append({"run": run, "rank": r, "step": t, "row": int(i), "octave": int(got["kps"]["octave"][i]), "bits": b})
    return bad


while time.time() - t_start < args.seconds and runs < args.max_runs:
    results = [[[None] * T for _ in range(world)] for _ in range(inner)]
    errors = []
    def rank_main(r):
        try:
            mine = shard_cameras(n_cams, world, r)
            fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=nf)], w, h, rank=r, world_size=world, global_cams=mine)
            if not args.no_exchange:
                fe.fe.exchange_init_loopback(3000 + runs, world, r)
                fe.native_exchange = True
                if not args.no_probe: fe.fe.debug_exchange_timing(True)
            for k in range(inner):
                if k: fe.reset()
                announced = 0
                for t in range(T):
                    if (r == 1 or args.all_delay) and not args.no_delay: time.sleep(1.0e-3)
                    while announced < min(t + ahead, T - 1):
                        announced += 1
                        fe.announce([frames[announced][g] for g in mine])
                    announced = max(announced, t)
                    res = fe.step([frames[t][g] for g in mine])
                    results[k][r][t] = {"kps": res["kps"].copy(), "desc": res["desc"].copy()}
                    if not args.no_exchange and not args.no_probe: fe.fe.debug_exchange_us()
            if not args.no_exchange: fe.fe.exchange_shutdown()
            fe.close()
        except Exception as e:
            errors.append((r, repr(e)))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join(300) for x in th]
    assert not errors, errors
    for k in range(inner):
        bad_runs += check_run(results[k], runs); runs += 1
el = time.time() - t_start
out = {"tag": args.tag, "lib": os.path.basename(_lib.LIB_PATH), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "world": world,
       "options": {"persistent": args.persistent, "no_probe": args.no_probe, "no_delay": args.no_delay, "no_exchange": args.no_exchange, "all_delay": args.all_delay, "placement": args.placement,
                   "MORB_CHAIN_GRAPH": os.environ.get("MORB_CHAIN_GRAPH"), "MORB_TOP2_MFMA": os.environ.get("MORB_TOP2_MFMA"),
                   "MORB_TOP2_FP4": os.environ.get("MORB_TOP2_FP4")},
       "bad_steps_histogram": np.bincount(np.array(steps_bad, int), minlength=T).tolist(),
       "runs": runs, "bad_runs": bad_runs, "seconds": round(el, 1), "s_per_run": round(el / max(runs, 1), 3),
       "wrong_rows": len(bits_seen), "lanes": sorted({b // 4 for bs in bits_seen for b in bs}), "js": [sorted({b % 4 for b in bs}) for bs in bits_seen][:40],
       "detail": bad_detail[:40]}
if check:
    c = read_check()
    out["selfcheck"] = {"bad_load": c[0], "bad_arith": c[1], "bad_lds": c[2], "bad_bit": c[3], "keypoints": c[4], "records": c[5],
                        "bad_arith_by_quad_j_and_coordinate(r0,c0,r1,c1)": [c[232 + 4 * j: 236 + 4 * j] for j in range(4)],
                        "bad_arith_by_lane_quarter": c[248:252]}
    recs = []
    for s in range(min(c[5], 16)):
        rw = c[8 + 14 * s: 8 + 14 * (s + 1)]
        f = lambda v: float(np.array([v & 0xffffffff], np.uint32).view(np.float32)[0])
        i = lambda v: int(np.array([v & 0xffffffff], np.uint32).view(np.int32)[0])
        recs.append({"ki": rw[0] >> 32, "lane": (rw[0] >> 8) & 0xff, "j": (rw[0] >> 4) & 0xf, "flags(load,arith,lds,bit)": rw[0] & 0xf,
                     "q_used": [f(rw[1] >> 32), f(rw[1]), f(rw[2] >> 32), f(rw[2])], "q_mem": [f(rw[3] >> 32), f(rw[3]), f(rw[4] >> 32), f(rw[4])],
                     "a_b": [f(rw[5] >> 32), f(rw[5])], "rc_used": [i(rw[6] >> 32), i(rw[6]), i(rw[7] >> 32), i(rw[7])],
                     "rc_again": [i(rw[8] >> 32), i(rw[8]), i(rw[9] >> 32), i(rw[9])], "t_used": [i(rw[10] >> 32), i(rw[10])],
                     "t_again": [i(rw[11] >> 32), i(rw[11])], "clock": rw[12], "block": rw[13] >> 32, "camlevel": rw[13] & 0xffffffff})
    out["records"] = recs
line = json.dumps(out)
print(line, flush=True)
if args.out:
    with open(args.out, "a") as f: f.write(line + "\n")
