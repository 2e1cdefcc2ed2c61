#!/usr/bin/env python3
"""The slow-peer loopback scenario of tests/test_gpu_frontend.py run N times; on a mismatch against the oracle prints WHAT differs:
rank, step, rows, bytes and bits per row."""
import os, sys, threading, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, synth
from multi_orb_slam_amd.dist import shard_cameras
from oracle_pipeline import OracleFrontEnd
os.environ["MORB_EXCHANGE_PLACEMENT"] = sys.argv[2] if len(sys.argv) > 2 else "inline"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
world, n_cams, w, h, nf, ahead, T = 4, 4, 640, 480, 1000, 2, 10
frames = [{g: synth.image(g, t, w, h) for g in range(n_cams)} for t in range(T)]
ofes = [OracleFrontEnd([m.ExtractorParams(nfeatures=nf)], w, h, shard_cameras(n_cams, world, r)) for r in range(world)]
exp_desc = None
bad_runs = 0
for run in range(N):
    results = [[None] * T for _ in range(world)]
    errors = []
    def rank_main(r):
        try:
            mine = shard_cameras(n_cams, world, r)
            fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=nf)], w, h, rank=r, world_size=world, global_cams=mine)
            fe.fe.exchange_init_loopback(3000 + run, world, r)
            fe.native_exchange = True
            fe.fe.debug_exchange_timing(True)
            announced = 0
            for t in range(T):
                if r == 1: time.sleep(1.0e-3)
                while announced < min(t + ahead, T - 1):
                    announced += 1
                    fe.announce([frames[announced][g] for g in mine])
                announced = max(announced, t)
                results[r][t] = fe.step([frames[t][g] for g in mine])
                fe.fe.debug_exchange_us()
            fe.fe.exchange_shutdown(); fe.close()
        except Exception as e:
            errors.append((r, repr(e)))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join(120) for x in th]
    assert not errors, errors
    if exp_desc is None:   # the extraction part of the oracle once: (kps, desc) per (rank, step)
        exp_desc = [[None] * T for _ in range(world)]
        import oracle
        for r in range(world):
            for t in range(T):
                exp_desc[r][t] = oracle.extract(frames[t][shard_cameras(n_cams, world, r)[0]], nfeatures=nf)
    bad = False
    for r in range(world):
        for t in range(T):
            got = results[r][t]
            ek, ed = exp_desc[r][t]
            if len(ed) != len(got["desc"]) or got["kps"].tobytes() != ek.tobytes():
                print("run", run, "rank", r, "step", t, "KEYPOINTS differ", len(ed), len(got["desc"])); bad = True; continue
            diff = got["desc"] != ed
            if diff.any():
                rows = np.flatnonzero(diff.any(axis=1))
                bits = [np.flatnonzero(np.unpackbits(got["desc"][i] ^ ed[i], bitorder="little")).tolist() for i in rows[:12]]
                print("run", run, "rank", r, "step", t, "DESC rows differ:", len(rows), "of", len(ed), "first rows", rows[:12].tolist(), "bits", bits,
                      "octaves", got["kps"]["octave"][rows[:12]].tolist(), "x", got["kps"]["x"][rows[:6]].tolist(), "y", got["kps"]["y"][rows[:6]].tolist(), flush=True)
                bad = True
    bad_runs += bad
print("runs with a mismatch: %d of %d" % (bad_runs, N))
