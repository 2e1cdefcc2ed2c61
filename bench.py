#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB front end (N-camera extract + match) on MI355X, with the matcher roofline and the
CPU baseline in the same run.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

One step = one timestep of one 2-camera 640x480 rig (BASELINE.json configs[1]): extract both cameras (8-level
pyramid, 1000 features/camera), merge the frame, SearchByProjection of the previous frame's points, exhaustive
cross-camera Hamming top-2.  Inputs are synthetic, generated once and resident in HBM before the timed region.
Every rank owns one rig (weak scaling); with N > 1 the cross-camera matcher sees every rank's descriptors through
one RCCL all-gather per step.  `value` = rig-frames/s summed over ranks.

Before anything is timed the GPU results of three steps are compared bit-for-bit with the CPU oracle.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, NFEAT, CAMS_PER_RANK, RING = 640, 480, 1000, 2, 8
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MATRIX_N = 32000                 # all-pairs size of configs[4]: 8 cameras x 4000 descriptors


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-overlap", action="store_true", help="every step extracts its own images first (no orbf_prefetch)")
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--matrix-n", type=int, default=MATRIX_N)
    return ap.parse_args()


def matcher_roofline(rt, m, stream, n, iters=80):
    """`roofline` of the bench line: the distance matrix in its default (matrix-core) form, with the xor/popcount form of
    the same kernel -- the formulation north_star names -- timed beside it under `popcount_form`.
    80 launches (~35 ms): the chip boosts for the first ~6 launches (~410 us), dips for the next dozen (~510 us) and then
    settles (profiles/r01/notes_experiments.md); the average over a run this long is the sustained figure."""
    out = _matrix_launches(rt, m, stream, n, iters)
    prev = m.Matcher.use_matrix_cores(0)
    try:
        alt = _matrix_launches(rt, m, stream, n, max(5, iters // 4))
    finally:
        m.Matcher.use_matrix_cores(prev)
    out["popcount_form"] = {"kernel": "k_hamming_matrix", "achieved": alt["achieved"], "frac": alt["frac"],
                            "avg_launch_us": alt["avg_launch_us"]}
    return out


def _matrix_launches(rt, m, stream, n, iters):
    """Hamming distance-matrix kernel (k_hamming_matrix_mfma at this size), Q = R = n: algorithmic bytes 32(Q+R) + 2QR per launch,
    average launch duration from HIP events on the stream the kernel runs on."""
    from multi_orb_slam_amd import synth
    d = synth.descriptors(n, 4242)
    dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dout = rt.DeviceBuffer(n * n * 2)
    dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
    for _ in range(3):
        m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, stream)
    rt.stream_sync(stream)
    e0, e1 = rt.Event(), rt.Event()
    e0.record(stream)
    for _ in range(iters):
        m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, stream)
    e1.record(stream)
    ms = e0.elapsed_ms(e1) / iters
    alg_bytes = 32.0 * (n + n) + 2.0 * n * n
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    # spot-check of the timed buffer against the host popcount (rows 0 and n-1)
    import numpy as np
    row = dout.download(np.uint16, n, stream, offset=(n - 1) * n * 2)
    ref = np.unpackbits(d[n - 1][None, :] ^ synth.perturbed_queries(d, 9), axis=1).sum(1).astype(np.uint16)
    assert np.array_equal(row, ref), "distance matrix spot check failed"
    traffic = None
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pj):
        try:
            pm = json.load(open(pj))
            traffic = (pm.get("k_hamming_matrix_mfma") or pm.get("k_hamming_matrix") or {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    for b in (dq, dr, dout):
        b.free()
    return {"kernel": "k_hamming_matrix_mfma", "workload": "Q=R=%d uint16 distance matrix" % n, "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "alg_bytes_per_launch": alg_bytes, "avg_launch_us": round(ms * 1e3, 2)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or os.environ.get("MORB_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on one GPU
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ.setdefault("MASTER_PORT", "29511")
        # (no device_id=: eager communicator init made every hipStreamSynchronize of this process ~100 us slower here)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    import numpy as np
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import synth, pipeline, rt
    from multi_orb_slam_amd.dist import DescriptorExchange

    rt.set_device(local)
    params = [m.ExtractorParams(nfeatures=NFEAT)] * CAMS_PER_RANK
    gcam = [rank * CAMS_PER_RANK + c for c in range(CAMS_PER_RANK)]
    fe = pipeline.FrontEnd(params, W, H, device=local, rank=rank, world_size=world, global_cams=gcam)
    if use_dist:
        import torch
        fe.gather = DescriptorExchange(torch.device("cuda", local), dist)
        fe.world = max(world, 2) if world == 1 else world   # world 1 + forced exchange still takes the block path
        # the all-gather from inside the native step (RCCL's C API; torch.distributed ships the communicator id once);
        # MORB_NATIVE_EXCHANGE=0 keeps the torch.distributed collective of DescriptorExchange
        if os.environ.get("MORB_NATIVE_EXCHANGE", "1") != "0":
            fe.enable_native_exchange(dist, torch.device("cuda", local))

    # ---- synthetic stream of this rank's rig, resident in HBM before timing
    host_frames = [[synth.image(g, t, W, H) for g in gcam] for t in range(RING)]
    dev_frames = []
    for t in range(RING):
        row = []
        for c in range(CAMS_PER_RANK):
            b = rt.DeviceBuffer(W * H); b.upload(host_frames[t][c]); row.append(b)
        dev_frames.append(row)
    rt.device_sync()

    def frame_args(t):
        return [(dev_frames[t % RING][c].ptr, W) for c in range(CAMS_PER_RANK)]

    # Consecutive timesteps overlap: while step t is matched (and, with N > 1, exchanged), the extraction of step t+1 already
    # runs on the extractor's stream.
    overlap = not a.no_overlap

    # ---- parity gate: four steps bit-exact vs the CPU oracle (single-rank view; N > 1 checks its own cameras)
    parity = "skipped"
    if world == 1:
        from oracle_pipeline import OracleFrontEnd, assert_same_step
        ofe = OracleFrontEnd(params, W, H, gcam)
        if overlap:
            fe.announce(frame_args(1), resident=True)
        for t in range(5):   # same call pattern as the timed loop: two future steps are announced (orbf_prefetch)
            got = fe.step(frame_args(t), resident=True, next_images=frame_args(t + 2) if overlap else None)
            assert_same_step(got, ofe.step(host_frames[t % RING]))
        parity = "bit-exact vs oracle on 5 steps (keypoints, descriptors, temporal + cross-camera matches)"
        fe.reset()

    def sync_all():
        rt.device_sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier(device_ids=[local])

    ahead = [0]   # index of the youngest timestep announced so far

    def run(nsteps, t0, overlap=overlap):
        # every step completes one timestep (extract + match); with `overlap` the images of the two steps after it are known
        # to the front end (one new announcement per step), so K steps enqueue K extractions and complete K matchings
        if overlap and ahead[0] < t0 + 1:
            fe.announce(frame_args(t0 + 1), resident=True); ahead[0] = t0 + 1
        for i in range(nsteps):
            if overlap:
                ahead[0] = t0 + i + 2
            fe.step(frame_args(t0 + i), resident=True, next_images=frame_args(t0 + i + 2) if overlap else None)

    fe.copy_results = False          # timed loop: consume the results in place (views of the pinned buffers)
    # The interpreter's cyclic collector would otherwise run inside the loop (every torch.distributed call allocates
    # containers; one young-generation pass costs ~120 us with torch imported): park it for the timed region.
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    run(a.warmup, 0)
    sync_all()
    t_start = time.perf_counter()
    run(a.steps, a.warmup)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = world * a.steps / elapsed
    # for reference: the same loop with every step extracting its own images first (latency of one isolated timestep)
    serial_ms = None
    if overlap:
        fe.reset(); run(20, 0, False); sync_all()
        t1 = time.perf_counter(); run(200, 20, False); sync_all()
        serial_ms = 1e3 * (time.perf_counter() - t1) / 200
        fe.reset(); ahead[0] = 0

    gc.enable(); gc.unfreeze()
    # per-stage GPU time of the extractor (HIP events) on one extra profiled step
    fe.ex.set_profiling(True)
    fe.step(frame_args(0), resident=True); fe.step(frame_args(1), resident=True)
    stages = fe.ex.stage_times_us()
    fe.ex.set_profiling(False)

    out = {
        "metric": "frames/sec (N-cam extract+match)", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: one 2-cam 640x480 rig per GPU, 8-level pyramid, 1000 feat/cam, HIP "
                               "FAST+rBRIEF extract + SearchByProjection + cross-camera Hamming top-2",
                   "cams_per_gpu": CAMS_PER_RANK, "width": W, "height": H, "nfeatures": NFEAT, "nlevels": 8,
                   "frame_unit": "one rig timestep (2 cameras)"},
        "parity": parity,
        "overlap": ("the extractions of timesteps t+1 and t+2 run next to the matching of timestep t (orbf_prefetch, two "
                    "extractor instances); one isolated timestep takes %.4f ms" % serial_ms) if overlap else "off",
        "extractor_stage_us": {k: round(v, 1) for k, v in stages.items()},
        "exchange": ("none (one rank)" if not use_dist else
                     "one RCCL all-gather of the step's descriptor block per step, issued natively from inside the step (RCCL C API)"
                     if getattr(fe, "native_exchange", False) else
                     "one RCCL all-gather of the step's descriptor block per step through torch.distributed"),
    }
    if rank == 0 and not a.no_roofline:
        out["roofline"] = matcher_roofline(rt, m, fe.stream, a.matrix_n)
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle_pipeline import OracleFrontEnd

        def cpu_rate(cam_threads):
            ofe = OracleFrontEnd(params, W, H, gcam, cam_threads=cam_threads)
            ofe.step(host_frames[0])                  # warm caches, establish `prev`
            t0 = time.perf_counter()
            for i in range(a.cpu_frames):
                ofe.step(host_frames[(1 + i) % RING])
            return a.cpu_frames / (time.perf_counter() - t0)

        v1 = cpu_rate(False)      # faithful to the reference: cameras back to back on the tracking thread (src/Frame.cc:182,185)
        vn = cpu_rate(True)       # one thread per camera (the variant commented out at src/Frame.cc:106-109)
        out["cpu_baseline"] = {"value": round(v1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d steps of the same 2-cam 640x480 workload through oracle/liborb_oracle.so "
                                         "(scalar C++ restatement, 1 thread; host has %d cores)" % (a.cpu_frames, os.cpu_count()),
                               "value_one_thread_per_camera": round(vn, 3), "cores_one_thread_per_camera": CAMS_PER_RANK}
    if rank == 0:
        print(json.dumps(out))
    fe.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
