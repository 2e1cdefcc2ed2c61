#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB front end (N-camera extract + match) on MI355X, with the matcher rooflines and the
CPU baseline in the same run.

  python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4]
  (N > 1: one rank per GPU over RCCL -- launched by torch.distributed.run, or, when run as a plain process, by bench.py
  itself: the parent starts N rank processes before it touches a GPU and relays rank 0's line)

One step = one timestep of one camera rig: extract every camera (8-level pyramid), merge the frame, SearchByProjection of
the previous frame's points, exhaustive cross-camera Hamming top-2.  Inputs are synthetic, generated once and resident in
HBM before the timed region.  --config selects the BASELINE.json workload:

  1 (default)  configs[1]  2 x 640x480 @1000    every rank owns one whole rig                       weak scaling
  2            configs[2]  2 x 1280x720 @2000   every rank owns one whole rig                       weak scaling
  3            configs[3]  4 x 640x480 @1000    ONE rig, cameras sharded over the ranks (4/N each)  strong scaling
  4            configs[4]  8 x 1920x1080 @4000  ONE rig, cameras sharded over the ranks (8/N each)  strong scaling

With N > 1 the cross-camera matcher sees every rank's descriptors through ONE RCCL all-gather per step, issued natively
from inside the step.  `value` = rig timesteps per second of the whole job (weak: summed over the ranks' rigs).

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + device synchronisation, repeated until
at least --min-time seconds have been timed (so that a short K still gives a stable figure); `ms_per_step` is the median
block, per-step median / p5 / p95 come from host timestamps of the individual step calls (each ends with the step's single
synchronisation).  Before anything is timed the GPU results are compared bit for bit with the CPU oracle.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
INT_VALU_PEAK_TOPS = 39.3        # 256 CU x 4 SIMD x 16 int lanes/clk x 2.4 GHz (measured 36.9 for v_xor/v_bcnt/v_add)
I8_MFMA_PEAK_TOPS = 5000.0       # dense int8 matrix peak = 2 x the 2.5 PF bf16 dense peak (microbench ceiling in the guide: 3944)
FP4_MFMA_PEAK_TOPS = 10000.0     # dense FP4 / FP6 matrix peak (MI355X_MICROARCH.md: ~10 PF dense; measured 9 099 with 32x32x64)
MATRIX_N = 32000                 # all-pairs size of configs[4]: 8 cameras x 4000 descriptors

CONFIGS = {
    1: dict(name="configs[1]", width=640, height=480, nfeatures=1000, rig_cams=2, scaling="weak", ring=8,
            text="one 2-cam 640x480 rig per GPU, 8-level pyramid, 1000 feat/cam"),
    2: dict(name="configs[2]", width=1280, height=720, nfeatures=2000, rig_cams=2, scaling="weak", ring=8,
            text="one 2-cam 1280x720 rig per GPU, 8-level pyramid, 2000 feat/cam, radius-gated SearchByProjection"),
    3: dict(name="configs[3]", width=640, height=480, nfeatures=1000, rig_cams=4, scaling="strong", ring=8,
            text="ONE 4-cam 640x480 rig, cameras sharded over the GPUs (one camera per GPU at N=4), 1000 feat/cam, "
                 "RCCL all-gather of the 256-bit descriptors for cross-camera matching"),
    4: dict(name="configs[4]", width=1920, height=1080, nfeatures=4000, rig_cams=8, scaling="strong", ring=4,
            text="ONE rig of 8 synthetic 1920x1080 streams, cameras sharded over the GPUs, 4000 feat/cam, batched pyramid "
                 "extract + all-pairs Hamming top-2 (32k x 28k at N=1)"),
}


def plan_for(config, world, rank):
    """Which cameras rank `rank` of `world` owns under --config `config`, and how the job's value is counted.  Pure
    arithmetic (no GPU, no torch): tests/test_bench_plan.py runs it for the torchrun shapes the driver uses."""
    if config not in CONFIGS:
        raise SystemExit("--config must be one of %s" % sorted(CONFIGS))
    c = dict(CONFIGS[config])
    if not (0 <= rank < world):
        raise SystemExit("rank %d outside world %d" % (rank, world))
    if c["scaling"] == "weak":
        cams = c["rig_cams"]
        gcam = [rank * cams + k for k in range(cams)]      # every rank: a rig of its own (distinct synthetic cameras)
        rigs = world
    else:
        if world > c["rig_cams"] or c["rig_cams"] % world:
            raise SystemExit("%s shards %d cameras: --gpus must divide %d (got %d)" % (c["name"], c["rig_cams"], c["rig_cams"], world))
        from multi_orb_slam_amd.dist import shard_cameras
        gcam = shard_cameras(c["rig_cams"], world, rank)
        rigs = 1
    c.update(cams_per_rank=len(gcam), global_cams=gcam, rigs=rigs, world=world, rank=rank,
             exchange=world > 1)
    return c


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs = rank processes.  Under a launcher (WORLD_SIZE set) it must equal the "
                    "world size; on its own `--gpus N` with N > 1 starts the N ranks itself.  Default: WORLD_SIZE, else 1")
    ap.add_argument("--plan-only", action="store_true", help="no GPU: launch / rendezvous (gloo) / sharding plan only, one JSON line")
    ap.add_argument("--steps", type=int, default=None, help="steps per timed block (default 2000; 100 for --config 4)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--min-time", type=float, default=0.25, help="keep timing blocks of --steps steps until this many seconds are covered")
    ap.add_argument("--no-overlap", action="store_true", help="every step extracts its own images first (no orbf_prefetch)")
    ap.add_argument("--ahead", type=int, default=3, choices=[1, 2, 3],
                    help="timesteps whose images the front end knows ahead of the step it matches (orbf_prefetch); default 3")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the stored counter passes (profiles/pmc_traffic.json) instead of two rocprofv3 --pmc passes run now")
    ap.add_argument("--matrix-n", type=int, default=MATRIX_N)
    a = ap.parse_args(argv)
    if a.steps is None:
        a.steps = 100 if a.config == 4 else 2000
    if a.warmup is None:
        a.warmup = 10 if a.config == 4 else 200
    return a


def pct(xs, p):
    """p-th percentile (nearest rank) of a non-empty list"""
    s = sorted(xs)
    return s[min(len(s) - 1, max(0, int(math.ceil(p / 100.0 * len(s))) - 1))]


# ------------------------------------------------------------------------------------------------ matcher rooflines
def _settled_launches(rt, run, stream, iters, group=10, tol=0.003, max_groups=40):
    """Average duration (ms) of `iters` launches of `run` AFTER the launch duration has settled, + how many launches that took.
    A kernel that keeps the whole chip busy runs its first ~100-150 launches (tens of ms after an idle period) up to 20 %
    slower than the ones after -- the clocks take that long to settle (the curve is in
    profiles/r03/notes_experiments.md) -- so the sustained figure is timed only once the mean of three consecutive groups of
    `group` launches is within `tol` of the mean of the three groups before (or after `max_groups` groups).  HIP events on the stream the kernels are launched on."""
    ev = [rt.Event() for _ in range(max_groups + 1)]
    ev[0].record(stream)
    hist = []
    used = 0
    for g in range(max_groups):
        for _ in range(group):
            run()
        ev[g + 1].record(stream)
        hist.append(ev[g].elapsed_ms(ev[g + 1]) / group)   # (waits for the group: the transient is tens of groups long)
        used += group
        if len(hist) >= 6 and abs(sum(hist[-3:]) - sum(hist[-6:-3])) <= tol * sum(hist[-3:]):   # (no drift left between 3-group means)
            break
    e0, e1 = rt.Event(), rt.Event()
    e0.record(stream)
    for _ in range(iters):
        run()
    e1.record(stream)
    return e0.elapsed_ms(e1) / iters, used, [round(h * 1e3, 1) for h in hist]


def device_state():
    """Clocks, power and temperature as rocm-smi reports them right now (the distance matrix runs in one of two bands
    per process, 366-368 / 397-401 us, that nothing in the kernel explains -- whatever the part reports at roofline time goes into
    the line so that the bands can be held against it).  None where the tool is missing or says nothing."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
    except Exception:
        return None
    keep = {}
    for k, v in card.items():
        lk = k.lower()
        if any(w in lk for w in ("sclk", "mclk", "fclk", "socclk", "power", "temperature (sensor junction)", "temperature (sensor memory)", "performance level")):
            keep[k] = v
    return keep or None


def matcher_roofline(rt, m, stream, n, iters=200):
    """`roofline` of the bench line (M2): the uint16 distance matrix in its default (matrix-core) form, with the xor/popcount
    form of the same kernel -- the formulation north_star names -- timed beside it under `popcount_form`.
    200 launches (~80 ms) timed after the launch duration has settled (_settled_launches): the sustained figure."""
    before = device_state()
    out = _matrix_launches(rt, m, stream, n, iters)
    out["device_state"] = {"before": before, "after": device_state()}
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
    import numpy as np
    from multi_orb_slam_amd import synth
    d = synth.descriptors(n, 4242)
    dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dout = rt.DeviceBuffer(n * n * 2)
    dq.upload(d); dr.upload(synth.perturbed_queries(d, 9))
    ms, settle_launches, settle_curve = _settled_launches(
        rt, lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, stream), stream, iters)
    alg_bytes = 32.0 * (n + n) + 2.0 * n * n
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    # spot-check of the timed buffer against the host popcount (row n-1)
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
            "traffic": traffic, "alg_bytes_per_launch": alg_bytes, "avg_launch_us": round(ms * 1e3, 2), "timed_launches": iters,
            "launches_before_timing": settle_launches, "settling_us_per_launch_groups_of_10": settle_curve}


def top2_roofline(rt, m, stream, n, iters=200):
    """`roofline_m1`: exhaustive top-2, Q = R = n.  18 int ops per 256-bit pair (8 xor + 8 popcount +
    2 compare/select) against the integer vector peak for the xor/popcount form; the default matrix-core form of the same
    entry point (int8 dot product of the +-1-expanded descriptors, 512 int8 ops per pair) against the dense int8 MFMA peak."""
    import numpy as np
    from multi_orb_slam_amd import synth
    d = synth.descriptors(n, 777)
    qh = synth.perturbed_queries(d, 11)
    dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32)
    dq.upload(qh); dr.upload(d)
    o = [rt.DeviceBuffer(n * 4) for _ in range(3)]
    out = {}
    pairs = float(n) * n
    for form, on in (("mfma_form", 1), ("int8_form", 2), ("popcount_form", 0)):
        prev = m.Matcher.use_matrix_cores(1 if on else 0)
        prev_fp4 = m.Matcher.use_fp4_top2(0 if on == 2 else -1)
        try:
            sb = m.Matcher.top2_scratch_bytes(n, n)
            scratch = rt.DeviceBuffer(max(sb, 16))
            run = lambda: m.Matcher.hamming_top2_device(dq.ptr, n, dr.ptr, n, o[0].ptr, o[1].ptr, o[2].ptr, scratch.ptr if sb else None, stream)
            ms, settle_launches, settle_curve = _settled_launches(rt, run, stream, iters if on == 1 else max(5, iters // 4))
            bi = o[0].download(np.int32, 64, stream); bd = o[1].download(np.int32, 64, stream)
            for i in range(0, 64, 9):   # spot check against the host popcount
                dist = np.unpackbits(d ^ qh[i], axis=1).sum(1)
                assert bd[i] == dist.min() and bi[i] == int(np.argmin(dist)), "top-2 spot check failed"
            scratch.free()
        finally:
            m.Matcher.use_matrix_cores(prev); m.Matcher.use_fp4_top2(prev_fp4)
        if on:
            # 256 multiply-adds = 512 operations per pair either way; the FP4 form runs them on v_mfma_f32_32x32x64_f8f6f4 (dense
            # FP4 peak 10 PFLOP/s), the int8 form on v_mfma_i32_32x32x32_i8 (5 POP/s): each is priced against ITS instruction's peak
            ach = 512.0 * pairs / (ms * 1e-3) / 1e12
            peak = FP4_MFMA_PEAK_TOPS if on == 1 else I8_MFMA_PEAK_TOPS
            out[form] = {"kernel": "k_hamming_top2_mfma<fp4>" if on == 1 else "k_hamming_top2_mfma<int8>", "bound": "mfma",
                         "achieved": round(ach, 1), "peak": peak, "unit": "fp4 TOP/s" if on == 1 else "int8 TOP/s",
                         "frac": round(ach / peak, 4), "avg_launch_us": round(ms * 1e3, 2),
                         "pairs_per_s": round(pairs / (ms * 1e-3), 0), "timed_launches": iters if on == 1 else max(5, iters // 4),
                         "launches_before_timing": settle_launches, "settling_us_per_launch_groups_of_10": settle_curve}
            if on == 1:
                out[form]["note"] = ("the vector ALU (two instructions per key + the bit -> FP4 expansion) is this form's limit, not the matrix "
                                     "pipe: %.2f of the FP4 peak is %.2f of what the int8 instruction could deliver at its peak"
                                     % (ach / peak, ach / I8_MFMA_PEAK_TOPS))
        else:
            ach = 18.0 * pairs / (ms * 1e-3) / 1e12
            out[form] = {"kernel": "k_hamming_top2", "bound": "valu", "achieved": round(ach, 2), "peak": INT_VALU_PEAK_TOPS,
                         "unit": "int32 T lane-op/s", "frac": round(ach / INT_VALU_PEAK_TOPS, 4), "avg_launch_us": round(ms * 1e3, 2),
                         "pairs_per_s": round(pairs / (ms * 1e-3), 0)}
    for b in [dq, dr] + o:
        b.free()
    out["workload"] = "Q=R=%d exhaustive top-2; algorithmic 18 int ops / pair (8 xor + 8 popcount + 2 select), bytes 32(Q+R)+12Q" % n
    return out


def project_roofline(m, n_feat=2000, width=1280, height=720, iters=50):
    """`roofline_m3`: the projection-gated kernel alone on configs[2]'s matcher workload: a frame of
    2 x n_feat extracted features, every feature of the previous timestep projected into it.  Algorithmic bytes per query:
    68 (the query record) + 8 per grid cell of its window + per gated candidate 4 (index) + 16 (x, y, octave, uright) +
    32 (descriptor) + 52 B of shortlist / count output; achieved = bytes / HIP-event launch time."""
    import numpy as np
    from multi_orb_slam_amd import synth, pipeline
    import multi_orb_slam_amd as mm
    params = [mm.ExtractorParams(nfeatures=n_feat)] * 2
    ex = mm.Extractor(params, width, height)
    fr = [ex.extract([synth.image(c, t, width, height) for c in range(2)]) for t in range(2)]
    ex.close()
    mt = mm.Matcher()
    data = mm.FrameData.from_cameras(fr[1], width, height)
    frame = mt.frame(data)
    k0 = np.concatenate([k for k, _ in fr[0]]); d0 = np.concatenate([d for _, d in fr[0]])
    cam_of = np.repeat(np.arange(2, dtype=np.int32), [len(k) for k, _ in fr[0]])
    q = pipeline.make_queries((k0, d0, np.full(len(k0), -1.0, np.float32), cam_of), mm.tables(params[0])["scale"])
    us, gated = mt.time_project(frame, q, 100, iters)
    r = q["radius"].astype(np.float64)
    cells = ((np.ceil((q["u"] + r) * 64.0 / width) - np.floor((q["u"] - r) * 64.0 / width) + 1).clip(1, 64) *
             (np.ceil((q["v"] + r) * 48.0 / height) - np.floor((q["v"] - r) * 48.0 / height) + 1).clip(1, 48)).sum()
    alg = 68.0 * len(q) + 8.0 * float(cells) + 52.0 * gated + 52.0 * len(q)
    frame.close(); mt.close()
    ach = alg / (us * 1e-6) / 1e9
    return {"kernel": "k_project", "workload": "configs[2] matcher: %d queries into a 2 x %d-feature %dx%d frame, %d gated candidates"
                      % (len(q), n_feat, width, height, gated),
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "alg_bytes_per_launch": alg, "avg_launch_us": round(us, 2),
            "note": "latency-bound gather (one wave per query, dependent cell -> item -> descriptor loads), not a bandwidth kernel"}


# ------------------------------------------------------------------------------------------------ self-launch
def launch_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start one rank process per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), relay rank 0's JSON line, exit with the ranks' status.
    This parent never touches the GPU (no HIP call, no torch import): the ranks are fresh child processes, nothing is
    re-executed in a process that has initialised a device."""
    import socket
    import subprocess
    import threading
    n = a.gpus
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = dict(os.environ)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this pool
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode("utf-8", "replace")); sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True); th.start()
    rc = 0
    live = set(range(n))
    stop_at = None          # when the surviving ranks were asked to terminate
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, code))
                for o in live:               # exactly the processes started above
                    procs[o].terminate()
                stop_at = time.time()
        if stop_at is not None and live and time.time() - stop_at > 10.0:
            # a rank that sits in a collective may ignore SIGTERM until its own watchdog fires: after a grace period the
            # children started above (plain child processes of this one, by their exact handles) are killed
            for o in live:
                procs[o].kill()
            stop_at = float("inf")
        time.sleep(0.05)
    th.join(5.0)
    return rc if rc >= 0 else 128 - rc


def plan_only(a, rank, world):
    """--plan-only: no GPU.  Every rank computes its shard, the ranks rendezvous over gloo and rank 0 prints the line the
    real run would head with (n_gpus, scaling, which rank owns which cameras) -- the launcher and the sharding arithmetic
    checked on a CPU box (tests/test_bench_plan.py)."""
    P = plan_for(a.config, world, rank)
    owned = [P["global_cams"]]
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        owned = [None] * world
        dist.all_gather_object(owned, P["global_cams"])
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "frames/sec (N-cam extract+match)", "value": None, "unit": "frames/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "scaling": P["scaling"], "plan_only": True,
                          "config": {"workload": "%s: %s" % (P["name"], P["text"]), "cams_per_gpu": P["cams_per_rank"],
                                     "rig_cams": P["rig_cams"], "rigs": P["rigs"]},
                          "cameras_of_rank": owned, "exchange": P["exchange"]}))
    return 0


def extract_alg_bytes(width, height, nfeatures, nlevels=8, scale=1.2):
    """algorithmic bytes of one extraction = W*H (level 0 read) + 2 * sum over levels >= 1 of w*h (each
    level is materialised: one quantised write, one read) + 60 B per feature (28-B keypoint + 32-B descriptor):
    1.65 MB (640x480 @1000) / 4.90 MB (1280x720 @2000) / 11.01 MB (1920x1080 @4000)."""
    import numpy as np
    b = float(width * height)
    s = np.float32(1.0)
    for _l in range(1, nlevels):
        s = np.float32(s * np.float32(scale))
        inv = np.float32(1.0) / s
        b += 2.0 * float(int(np.rint(np.float32(width) * inv))) * float(int(np.rint(np.float32(height) * inv)))
    return b + 60.0 * nfeatures


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (a.gpus or 1) > 1:
        return launch_ranks(a, argv)           # before any GPU call; the ranks are children of this process
    # a run that makes no progress ends with a traceback instead of holding its GPU box (a default run takes well under a minute)
    wd = int(os.environ.get("MORB_BENCH_WATCHDOG_S", "1500"))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    rank = int(os.environ.get("RANK", "0")); world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus is not None and a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a line "
                         "whose n_gpus is not the number of GPUs that worked" % (a.gpus, world))
    if a.plan_only:
        return plan_only(a, rank, world)
    dist = None; ctl = None
    use_dist = world > 1 or os.environ.get("MORB_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on one GPU
    json_out = sys.stdout
    if use_dist:
        # RCCL prints a version banner on the process' stdout when its first communicator comes up: keep fd 1 for the ONE
        # JSON line, send everything else any library writes there to stderr
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import torch
        import torch.distributed as dist
        # Transport of the per-step exchange: RCCL (one rank per GPU), or the peer transport -- direct writes into the other ranks'
        # IPC-mapped arenas -- which also works with SEVERAL RANKS PER GPU: a box with fewer GPUs than ranks (the 1-GPU lease) can
        # rehearse the whole N > 1 job that way (RCCL refuses two ranks on one device).  MORB_EXCHANGE_TRANSPORT = rccl | peer chooses;
        # default: peer exactly when the ranks have to share devices.
        ndev = max(1, torch.cuda.device_count())          # (counting devices does not initialise one)
        shared_devices = world > ndev
        transport = os.environ.get("MORB_EXCHANGE_TRANSPORT", "peer" if shared_devices else "rccl")
        if shared_devices and transport != "peer":
            raise SystemExit("bench.py: %d ranks on %d GPU(s) need MORB_EXCHANGE_TRANSPORT=peer (RCCL refuses two ranks on one device)" % (world, ndev))
        local = local % ndev
        torch.cuda.set_device(local)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ.setdefault("MASTER_PORT", "29511")
        if transport == "peer":
            # no RCCL at all: gloo carries the handles, the barriers and the MAX-reduced block times (host tensors)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ctl = dist.group.WORLD
        else:
            # (no device_id=: eager communicator init made every hipStreamSynchronize of this process ~100 us slower here)
            dist.init_process_group("nccl", rank=rank, world_size=world)
            # control plane: the long waits (other ranks idle while rank 0 times the CPU baseline and the matcher rooflines) sit on a
            # gloo barrier -- host sockets, no kernel spinning on the waiting GPUs
            ctl = dist.new_group(backend="gloo")
    import numpy as np
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import synth, pipeline, rt
    from multi_orb_slam_amd.dist import DescriptorExchange

    P = plan_for(a.config, world, rank)
    W, H, NFEAT, NC, RING, gcam = P["width"], P["height"], P["nfeatures"], P["cams_per_rank"], P["ring"], P["global_cams"]
    rt.set_device(local)
    params = [m.ExtractorParams(nfeatures=NFEAT)] * NC
    fe = pipeline.FrontEnd(params, W, H, device=local, rank=rank, world_size=world, global_cams=gcam)
    if use_dist:
        import torch
        fe.world = max(world, 2) if world == 1 else world   # world 1 + forced exchange still takes the block path
        if transport == "peer":
            if not fe.enable_peer_exchange(dist):
                raise SystemExit("bench.py: the peer transport could not be set up on every rank (hipIpc*; HSA_ENABLE_IPC_MODE_LEGACY=0?)")
        else:
            fe.gather = DescriptorExchange(torch.device("cuda", local), dist)
            # the all-gather from inside the native step (RCCL's C API; torch.distributed ships the communicator id once);
            # MORB_NATIVE_EXCHANGE=0 keeps the torch.distributed collective of DescriptorExchange
            if os.environ.get("MORB_NATIVE_EXCHANGE", "1") != "0":
                fe.enable_native_exchange(dist, torch.device("cuda", local))
        if getattr(fe, "native_exchange", False):
            seen = fe.fe.exchange_world
            assert seen == world, "RCCL communicator spans %d rank(s), the launcher started %d" % (seen, world)

    # ---- synthetic stream of this rank's cameras: resident in HBM before timing, and once more in page-locked host memory
    host_frames = [[synth.image(g, t, W, H) for g in gcam] for t in range(RING)]
    dev_frames, pin_frames = [], []
    for t in range(RING):
        row, prow = [], []
        for c in range(NC):
            b = rt.DeviceBuffer(W * H); b.upload(host_frames[t][c]); row.append(b)
            pb = rt.PinnedBuffer(W * H); pb.array[:] = host_frames[t][c].reshape(-1); prow.append(pb)
        dev_frames.append(row); pin_frames.append(prow)
    rt.device_sync()

    _prepared = {}

    def frame_args(t, pinned=False):
        # the ring's slots are marshalled once (pipeline.FrontEnd.prepare): the timed loop hands the binding finished orbf_image arrays
        key = (t % RING, pinned)
        arr = _prepared.get(key)
        if arr is None:
            src = pin_frames if pinned else dev_frames
            arr = _prepared[key] = fe.prepare([(src[t % RING][c].ptr, W) for c in range(NC)], "pinned" if pinned else True)
        return arr

    # Consecutive timesteps overlap: while step t is matched (and, with N > 1, exchanged), the extraction of step t+1 already
    # runs on the extractor's stream.
    overlap = not a.no_overlap

    def sync_all(long_wait=False):
        rt.device_sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            if long_wait or transport == "peer":
                dist.barrier(group=ctl)
            if transport != "peer":
                dist.barrier(device_ids=[local])

    # ---- parity gate, on EVERY rank of any world size: this rank's cameras bit for bit against the CPU oracle on the call
    # pattern of the timed loop (keypoint records, descriptors, stereo, undistorted positions, the temporal search with its
    # rotation histogram), and the rig-wide cross-camera top-2 of this rank's features against the oracle's brute force over
    # the descriptors of every OTHER camera of the job -- which the oracle extracts itself from the other ranks' synthetic
    # images (nothing a GPU produced goes into the expectation).  One host thread per camera (the oracle's C entry points
    # release the GIL).
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    job_cams = list(range(world * P["rig_cams"])) if P["scaling"] == "weak" else list(range(P["rig_cams"]))
    if not use_dist or world == 1:
        job_cams = list(gcam)
    other_g = [g for g in job_cams if g not in gcam]
    n_gate = 2 if a.config == 4 else 5
    t_gate = time.perf_counter()
    ofe = OracleFrontEnd(params, W, H, gcam, cam_threads=True)
    pool = ThreadPoolExecutor(max(1, min(len(other_g), 16))) if other_g else None
    AHEAD = max(1, min(a.ahead, fe.fe.ahead_depth))
    if overlap:
        for k in range(1, AHEAD):
            fe.announce(frame_args(k), resident=True)
    for t in range(n_gate):   # AHEAD future steps are announced (orbf_prefetch), as in the timed loop
        got = fe.step(frame_args(t), resident=True, next_images=frame_args(t + AHEAD) if overlap else None)
        desc_of = {}
        if other_g:
            ext = lambda g: oracle.extract(synth.image(g, t % RING, W, H), nfeatures=NFEAT)[1]
            desc_of = dict(zip(other_g, pool.map(ext, other_g)))
        seen_own = {}

        def others(c):
            return [desc_of[g] if g in desc_of else seen_own[g] for g in job_cams if g != gcam[c]]
        # the oracle front end extracts this rank's cameras first and then asks for the others per camera
        own_hook = lambda per_cam: seen_own.update({g: per_cam[i][1] for i, g in enumerate(gcam)})
        exp = ofe.step(host_frames[t % RING], other_descs=others if other_g else None, on_extracted=own_hook)
        assert_same_step(got, exp)
    if pool is not None:
        pool.shutdown()
    if hasattr(ofe, "pool") and ofe.pool is not None:
        ofe.pool.shutdown()
    parity = ("bit-exact vs oracle on %d steps, all %d camera(s) of this rank at full size (keypoints, descriptors, stereo, temporal "
              "matches incl. rotation histogram) + the complete cross-camera top-2 against the oracle's brute force over %d camera(s) "
              "of the job; every rank runs the gate (%.1f s on rank 0)" % (n_gate, NC, len(job_cams), time.perf_counter() - t_gate))
    fe.reset()
    sync_all(long_wait=True)      # a rank whose gate failed has raised by now: its exit stops the job

    ahead = [0]   # index of the youngest timestep announced so far
    stamps = []   # host timestamps after every step call of the current block
    native_us = []  # orbf_result::host_us of the same steps: time spent inside orbf_step_begin / _end, measured by the library

    def run(nsteps, t0, overlap=overlap, pinned=False, record=False):
        # every step completes one timestep (extract + match); with `overlap` the images of the two steps after it are known
        # to the front end (one new announcement per step), so K steps enqueue K extractions and complete K matchings
        res = "pinned" if pinned else True
        if overlap:
            for k in range(max(ahead[0] + 1, t0 + 1), t0 + AHEAD):
                fe.announce(frame_args(k, pinned), resident=res); ahead[0] = k
        pc = time.perf_counter
        for i in range(nsteps):
            if overlap:
                ahead[0] = t0 + i + AHEAD
            r = fe.step(frame_args(t0 + i, pinned), resident=res, next_images=frame_args(t0 + i + AHEAD, pinned) if overlap else None)
            if record:
                stamps.append(pc())
                if not overlap:              # (the library's own clock is only read for the isolated-latency figures)
                    native_us.append(r["host_us"])

    def timed_blocks(K, t0, min_time, overlap=overlap, pinned=False):
        """blocks of exactly K steps, barrier + synchronise on both sides of each; -> (block seconds [max over ranks], per-step s)"""
        blocks, per_step, covered, t = [], [], 0.0, t0
        n_blocks = None
        while True:
            sync_all()
            del stamps[:]
            t_start = time.perf_counter()
            run(K, t, overlap, pinned, record=True)
            # closing bracket: this rank's work has drained (synchronise), its clock is read, and the MAX over the ranks -- the
            # moment the slowest rank was done -- is the block's time; the reduction itself is the closing barrier and stays
            # outside the clock (a barrier's own latency is not the path's)
            rt.device_sync()
            if dist is not None:
                import torch
                torch.cuda.synchronize()
            el = time.perf_counter() - t_start
            if dist is not None:
                tt = torch.tensor([el], dtype=torch.float64, device="cpu" if transport == "peer" else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            blocks.append(el); covered += el; t += K
            prev = t_start
            for s in stamps:
                per_step.append(s - prev); prev = s
            if n_blocks is None:     # every rank derives the same count from the same (max-reduced) first block
                n_blocks = max(1, min(200, int(math.ceil(min_time / max(el, 1e-9)))))
            if len(blocks) >= n_blocks:
                return blocks, per_step, t

    def latency(ps, c_abi_from=None):
        o = {"median": round(1e3 * pct(ps, 50), 4), "p5": round(1e3 * pct(ps, 5), 4), "p95": round(1e3 * pct(ps, 95), 4), "steps": len(ps)}
        # the same steps as the C ABI sees them: begin (everything enqueued) + end (wait + collect), without the Python binding
        # around the two calls (argument marshalling, result views, the native count of accepted cross matches)
        c_abi = sorted(1e-3 * (h[1] + h[2] + h[3]) for h in (c_abi_from or []))
        if c_abi:
            o["c_abi_ms"] = {"median": round(pct(c_abi, 50), 4), "p5": round(pct(c_abi, 5), 4), "p95": round(pct(c_abi, 95), 4),
                             "what": "orbf_step_begin + orbf_step_end as timed inside the library (orbf_result::host_us[1..3])"}
        return o

    fe.copy_results = False          # timed loop: consume the results in place (views of the pinned buffers)
    # The interpreter's cyclic collector would otherwise run inside the loop (every torch.distributed call allocates
    # containers; one young-generation pass costs ~120 us with torch imported): park it for the timed region.
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    run(a.warmup, 0)
    blocks, per_step, t_next = timed_blocks(a.steps, a.warmup, a.min_time)
    med_block = pct(blocks, 50)
    rigs = P["rigs"]
    value = rigs * a.steps / med_block

    # ---- the same loop fed from page-locked HOST memory: the H2D copy of every image (PCIe) is inside the timed region
    h2d = None
    if world == 1:
        fe.reset(); ahead[0] = 0
        run(min(a.warmup, 50), 0, overlap, pinned=True)
        b2, ps2, _ = timed_blocks(a.steps, min(a.warmup, 50), a.min_time, overlap, pinned=True)
        h2d = {"value": round(rigs * a.steps / pct(b2, 50), 2), "ms_per_step": round(1e3 * pct(b2, 50) / a.steps, 4),
               "bytes_per_step": NC * W * H, "source": "page-locked host memory, hipMemcpy2DAsync inside the step"}
    # ---- the same overlapped loop inside the library (orbf_run_stream: prefetch + orbf_step_motion + the accepted-match count per
    # step, one native call per block): the stream as a C or C++ host would drive it, without the Python binding around every step
    c_abi_loop = None
    if world == 1:
        from multi_orb_slam_amd.matcher import TH_LOW
        fe.reset(); ahead[0] = 0
        ring = [[(dev_frames[t][c].ptr, W, H, W, 1) for c in range(NC)] for t in range(RING)]
        motion = (pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ)
        la = AHEAD if overlap else 0
        upto, t_loop = -1, 0
        _st, upto = fe.fe.run_stream(ring, t_loop, min(a.warmup, 50), la, upto, motion, TH_LOW, pipeline.BOW_RATIO); t_loop += min(a.warmup, 50)
        blk, covered = [], 0.0
        while True:
            sync_all()
            t_start = time.perf_counter()
            _st, upto = fe.fe.run_stream(ring, t_loop, a.steps, la, upto, motion, TH_LOW, pipeline.BOW_RATIO)
            sync_all()
            el = time.perf_counter() - t_start
            blk.append(el); covered += el; t_loop += a.steps
            if covered >= a.min_time or len(blk) >= 200:
                break
        c_abi_loop = {"value": round(rigs * a.steps / pct(blk, 50), 2), "ms_per_step": round(1e3 * pct(blk, 50) / a.steps, 4), "blocks": len(blk),
                      "what": "orbf_run_stream: the same steps (%d announced ahead, accepted-match count included) driven from inside the library" % la}
    # ---- one isolated timestep: no look-ahead, every step extracts its own images first (what a live rig sees as latency)
    fe.reset(); ahead[0] = 0
    n_iso = max(20, min(200, a.steps))
    run(min(20, a.warmup), 0, False)
    del native_us[:]
    _b3, ps3, _ = timed_blocks(n_iso, 20, min(a.min_time, 0.1), overlap=False)
    iso = latency(ps3, native_us)
    # ---- the live-rig number: the same isolated step with the images in page-locked HOST memory -- no look-ahead, the H2D
    # copies of all cameras inside the step (what Frame::Frame, src/Frame.cc:148-288, would wait for on a rig that delivers
    # host images)
    fe.reset(); ahead[0] = 0
    run(min(20, a.warmup), 0, False, pinned=True)
    del native_us[:]
    _b4, ps4, _ = timed_blocks(n_iso, 20, min(a.min_time, 0.1), overlap=False, pinned=True)
    iso_h2d = latency(ps4, native_us)
    iso_h2d["bytes_per_step"] = NC * W * H
    fe.reset(); ahead[0] = 0
    gc.enable(); gc.unfreeze()

    # per-stage GPU time of the extractor (HIP events): median over several profiled isolated steps (a single probe is noisy)
    fe.ex.set_profiling(True)
    probes = []
    for t in range(12):
        fe.step(frame_args(t), resident=True)
        if t >= 2:
            probes.append(fe.ex.stage_times_us())
    fe.ex.set_profiling(False)
    stages = {k: pct([p[k] for p in probes], 50) for k in probes[0]}
    ex_us = sum(stages[k] for k in ("pyramid", "fast_cells", "compact", "quadtree", "describe"))
    ex_bytes = NC * extract_alg_bytes(W, H, NFEAT)
    ex_ach = ex_bytes / (ex_us * 1e-6) / 1e9 if ex_us > 0 else 0.0

    # (ranks that share a device -- the peer transport's rehearsal on a box with fewer GPUs than ranks -- are not GPUs that worked)
    n_gpus_used = world if not (use_dist and world > 1 and shared_devices) else ndev
    out = {
        "metric": "frames/sec (N-cam extract+match)", "value": round(value, 2), "unit": "frames/s", "n_gpus": n_gpus_used, "n_ranks": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * med_block / a.steps, 4),
        "higher_is_better": True, "scaling": P["scaling"], "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: %s; HIP FAST+rBRIEF extract + SearchByProjection + cross-camera Hamming top-2" % (P["name"], P["text"]),
                   "cams_per_gpu": NC, "rig_cams": P["rig_cams"], "rigs": rigs, "width": W, "height": H, "nfeatures": NFEAT, "nlevels": 8,
                   "frame_unit": "one rig timestep (%d cameras)" % P["rig_cams"]},
        "timing": {"blocks": len(blocks), "steps_per_block": a.steps, "timed_s": round(sum(blocks), 4),
                   "block_ms_per_step": {"median": round(1e3 * med_block / a.steps, 4), "min": round(1e3 * min(blocks) / a.steps, 4),
                                         "max": round(1e3 * max(blocks) / a.steps, 4)},
                   "step_ms": {"median": round(1e3 * pct(per_step, 50), 4), "p5": round(1e3 * pct(per_step, 5), 4),
                               "p95": round(1e3 * pct(per_step, 95), 4), "n": len(per_step)}},
        "value_c_abi_loop": c_abi_loop["value"] if c_abi_loop else None, "c_abi_loop": c_abi_loop,
        "latency_ms_isolated": iso["median"], "latency_isolated": iso,
        "latency_ms_isolated_h2d": iso_h2d["median"], "latency_isolated_h2d": iso_h2d,
        "value_isolated": round(rigs * 1e3 / iso["median"], 2),
        "value_h2d_inclusive": h2d["value"] if h2d else None, "h2d_inclusive": h2d,
        "parity": parity,
        "ahead": AHEAD if overlap else 0,
        "overlap": ("the extractions of timesteps t+1 .. t+%d run next to the matching of timestep t (orbf_prefetch); `value` needs "
                    "the images %d steps ahead, `latency_ms_isolated` / `value_isolated` do not" % (AHEAD, AHEAD)) if overlap else "off",
        "extractor_stage_us": {k: round(v, 1) for k, v in stages.items()},
        "roofline_extract": {"kernel": "extraction chain (k_pyramid_tiled [k_ingest], k_fast_cells, k_octree, k_describe)", "bound": "hbm",
                             "achieved": round(ex_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ex_ach / HBM_PEAK_GBS, 5),
                             "alg_bytes_per_image": extract_alg_bytes(W, H, NFEAT), "images": NC, "chain_us": round(ex_us, 1),
                             "traffic": _pmc_extract_bytes(a.config), "traffic_unit": "HBM bytes per image of THIS configuration (stored counter passes)",
                             "note": "SURVEY 8(d): W*H + 2*sum(levels>=1) + 60*N per image over the GPU time of one isolated extraction "
                                     "chain (HIP events, median of 10 steps); a latency chain of small launches, not a bandwidth kernel"},
        "exchange": ("none (one rank)" if not use_dist else
                     ("peer transport: every rank writes its descriptor block straight into every other rank's IPC-mapped arena (one put kernel + "
                      "one wait kernel per step, no RCCL), issued at the tail of the step's extraction chain; %d rank process(es) on %d GPU(s)%s"
                      % (world, ndev, " -- ranks SHARE a device: a rehearsal of the N > 1 job, not a scaling measurement" if shared_devices else ""))
                     if transport == "peer" else
                     "one RCCL all-gather of the step's descriptor block per step, issued natively from inside the step (RCCL C API), %s"
                     % {1: "on the matcher's own stream behind the step's search (placement 1: inline)",
                        3: "at the tail of the step's extraction chain on the extractor's stream, steps ahead of its matching (placement 3: chain)"}.get(
                            fe.fe.exchange_placement, "placement unknown")
                     if getattr(fe, "native_exchange", False) else
                     "one RCCL all-gather of the step's descriptor block per step through torch.distributed"),
    }
    if rank == 0 and not a.no_roofline:
        out["roofline"] = matcher_roofline(rt, m, fe.stream, a.matrix_n)
        live = None if a.no_live_traffic else live_matrix_traffic(a.matrix_n)
        if live and live.get("hbm_bytes_per_launch"):
            out["roofline"]["traffic_stored"] = out["roofline"]["traffic"]
            out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = live
        else:
            out["roofline"]["traffic_source"] = {"source": "stored counter passes (profiles/pmc_traffic.json)", "live": live}
        out["roofline_m1"] = top2_roofline(rt, m, fe.stream, a.matrix_n)
        out["roofline_m3"] = project_roofline(m)
    if rank == 0 and world == 1 and not a.no_dropin and a.config == 1:
        import dropin_leg       # tests/dropin_leg.py: the C++ classes with the reference's signatures, reference call pattern
        out["dropin"] = dropin_leg.bench(W, H, NFEAT)
        out["dropin_fps"] = out["dropin"]["dropin_fps"]
    if rank == 0 and not a.no_cpu:
        # the CPU baseline of the unit `value` counts: one rig timestep.  Weak configurations: this rank's own rig; strong
        # ones: the WHOLE rig (all cameras, whichever rank owns them), so the figure is the same at every N.
        cpu_cams = list(gcam) if P["scaling"] == "weak" else list(range(P["rig_cams"]))
        cpu_params = [m.ExtractorParams(nfeatures=NFEAT)] * len(cpu_cams)
        cpu_frames = host_frames if cpu_cams == list(gcam) else [[synth.image(g, t, W, H) for g in cpu_cams] for t in range(min(RING, 3))]

        def cpu_rate(cam_threads, budget):
            o = OracleFrontEnd(cpu_params, W, H, cpu_cams, cam_threads=cam_threads)
            o.step(cpu_frames[0])                  # warm caches, establish `prev`
            t0 = time.perf_counter(); n = 0
            while n < 2 or (time.perf_counter() - t0 < budget and n < 400):
                o.step(cpu_frames[(1 + n) % len(cpu_frames)]); n += 1
            return n / (time.perf_counter() - t0), n

        v1, n1 = cpu_rate(False, 0.6 * a.cpu_seconds)   # faithful to the reference: cameras back to back on the tracking thread (src/Frame.cc:182,185)
        vn, _ = cpu_rate(True, 0.4 * a.cpu_seconds)     # one thread per camera (the variant commented out at src/Frame.cc:106-109)
        out["cpu_baseline"] = {"value": round(v1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d timesteps of one %d-cam %dx%d rig (the unit of `value`) through oracle/liborb_oracle.so "
                                         "(scalar C++ restatement, 1 thread; host has %d cores), timed on rank 0 while the other ranks wait"
                                         % (n1, len(cpu_cams), W, H, os.cpu_count()),
                               "value_one_thread_per_camera": round(vn, 3), "cores_one_thread_per_camera": len(cpu_cams),
                               "build": "g++ -O3 -ffp-contract=off (portable: the file built in the build container)"}
        # BASELINE.md quotes the CPU path at -O3 -march=native: the same sources built HERE for this host's CPU and timed in a
        # child process (the checker library of this process stays the portable one), output equality asserted through a digest
        out["cpu_baseline"]["march_native"] = cpu_native_leg(cpu_params, W, H, cpu_cams, cpu_frames, 0.5 * a.cpu_seconds)
    if rank == 0:
        json_out.write(json.dumps(out) + "\n"); json_out.flush()
    if dist is not None:
        dist.barrier(group=ctl)       # the other ranks wait here (host sockets) while rank 0 runs the rooflines and the CPU baseline
    fe.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def live_matrix_traffic(n):
    """`roofline.traffic` measured in THIS run: tools/matrix_once.py (1 GiB memset + 1 GiB copy as calibration, three launches of
    the distance-matrix kernel at Q = R = n) in a child process under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc
    WRITE_SIZE` (one counter per pass, nothing but the counters: MI355X_MICROARCH.md section HBM).  Both counters are in KiB;
    FETCH_SIZE counts 64 B per 128-B request on gfx950 (x2: the copy's ratio is reported).  None / a reason when the profiler is
    not there or a pass fails: the line then keeps the stored figure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"why": "rocprofv3 not found"}
    env = {k: v for k, v in os.environ.items() if not (k.startswith("ROCP") or k.startswith("ROCPROF") or k == "LD_PRELOAD")}
    env.update(TMPDIR="/tmp", MORB_NO_BAR_STAGING="1")
    GIB = float(1 << 30)
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            try:
                p = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(td, counter), "-o", "p", "--",
                                    sys.executable, os.path.join(ROOT, "tools", "matrix_once.py"), str(n)],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
                files = glob.glob(os.path.join(td, counter, "**", "*counter_collection.csv"), recursive=True)
                if p.returncode != 0 or not files:
                    return {"why": "%s pass failed (rc %s): %s" % (counter, p.returncode, (p.stderr or "")[-200:])}
                acc = {}
                for r in csv.DictReader(open(files[0])):
                    if r["Counter_Name"] != counter:
                        continue
                    for key in ("k_hamming_matrix_mfma", "fillBuffer", "copyBuffer"):
                        if key in r["Kernel_Name"]:
                            acc.setdefault(key, []).append(float(r["Counter_Value"]) * 1024.0)
                got[counter] = {k: sum(v) / len(v) for k, v in acc.items()}
            except Exception as e:      # noqa: BLE001 -- reported in the line
                return {"why": "%s pass: %r" % (counter, e)}
    try:
        fetch = got["FETCH_SIZE"]["k_hamming_matrix_mfma"] * 2.0
        write = got["WRITE_SIZE"]["k_hamming_matrix_mfma"]
        return {"source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) over tools/matrix_once.py",
                "hbm_bytes_per_launch": int(round(fetch + write)), "fetch_bytes_per_launch": int(round(fetch)),
                "write_bytes_per_launch": int(round(write)),
                "calibration": {"copy_1GiB_FETCH_SIZE_ratio": round(got["FETCH_SIZE"].get("copyBuffer", 0.0) / GIB, 4),
                                "memset_1GiB_WRITE_SIZE_ratio": round(got["WRITE_SIZE"].get("fillBuffer", 0.0) / GIB, 4),
                                "fetch_correction": 2.0}}
    except Exception as e:      # noqa: BLE001
        return {"why": "counters of the matrix kernel not found: %r" % (e,)}


def cpu_native_leg(cpu_params, W, H, cpu_cams, cpu_frames, budget):
    """`cpu_baseline.march_native`: oracle/_native/liborb_oracle.so (`make -C oracle native`: the oracle's sources with
    -march=native, built on this machine) timed on one thread over the same frames; the child loads ONLY that build
    (MORB_ORACLE_SO) and reports a digest of its first step, which must equal the portable build's."""
    import hashlib
    import subprocess
    import numpy as np
    odir = os.path.join(ROOT, "oracle")
    try:
        subprocess.check_call(["make", "-s", "-C", odir, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    except Exception as e:      # noqa: BLE001 -- reported in the line
        return {"value": None, "why": "could not build oracle/_native on this host: %r" % (e,)}
    import tempfile
    from oracle_pipeline import OracleFrontEnd

    def digest(r):
        h = hashlib.sha1()
        for k in ("kps", "desc", "uright", "match_of_feature"):
            h.update(np.ascontiguousarray(r[k]).tobytes())
        return h.hexdigest()
    o = OracleFrontEnd(cpu_params, W, H, cpu_cams)
    o.step(cpu_frames[0])
    want = digest(o.step(cpu_frames[1 % len(cpu_frames)]))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), frames=np.stack([np.stack(f) for f in cpu_frames]), cams=np.array(cpu_cams),
                 nf=np.array([p.nfeatures for p in cpu_params]), wh=np.array([W, H]), budget=np.array([budget]))
        code = (
            "import sys, os, json, time, hashlib\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "import multi_orb_slam_amd as m\n"
            "from oracle_pipeline import OracleFrontEnd\n"
            "z = np.load(sys.argv[1])\n"
            "frames = [list(f) for f in z['frames']]; W, H = [int(v) for v in z['wh']]\n"
            "params = [m.ExtractorParams(nfeatures=int(n)) for n in z['nf']]\n"
            "o = OracleFrontEnd(params, W, H, [int(c) for c in z['cams']])\n"
            "o.step(frames[0]); r = o.step(frames[1 %% len(frames)])\n"
            "h = hashlib.sha1()\n"
            "[h.update(np.ascontiguousarray(r[k]).tobytes()) for k in ('kps', 'desc', 'uright', 'match_of_feature')]\n"
            "t0 = time.perf_counter(); n = 0\n"
            "while n < 2 or (time.perf_counter() - t0 < float(z['budget'][0]) and n < 400):\n"
            "    o.step(frames[(2 + n) %% len(frames)]); n += 1\n"
            "print(json.dumps({'rate': n / (time.perf_counter() - t0), 'n': n, 'digest': h.hexdigest()}))\n"
        ) % (ROOT, os.path.join(ROOT, "tests"))
        env = dict(os.environ, MORB_ORACLE_SO=os.path.join(odir, "_native", "liborb_oracle.so"))
        try:
            p = subprocess.run([sys.executable, "-c", code, os.path.join(td, "in.npz")], env=env, capture_output=True, text=True, timeout=300)
            got = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001
            return {"value": None, "why": "child failed: %r" % (e,)}
    if got["digest"] != want:
        return {"value": None, "why": "the -march=native build's outputs differ from the portable build's (digest mismatch)"}
    return {"value": round(got["rate"], 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "build": "g++ -O3 -march=native -ffp-contract=off, built on this host (oracle/Makefile: native)",
            "sample": "%d timesteps, one thread, same frames; first step's outputs digest-equal to the portable build's" % got["n"]}


def _pmc_extract_bytes(config):
    """Counter-measured HBM bytes per image of the extraction chain for the configuration being run (profiles/pmc_traffic.json,
    section `extractor`: one FETCH_SIZE + WRITE_SIZE collection per image size; configs[3] is four cameras of configs[1]'s size)."""
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        sec = json.load(open(pj))["extractor"]["configs[%d]" % {1: 1, 2: 2, 3: 1, 4: 4}[config]]
        return sec["extract_chain"]["hbm_bytes_per_image"]
    except Exception:
        return None


def _pmc_bytes(key):
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return (json.load(open(pj)).get(key) or {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


if __name__ == "__main__":
    sys.exit(main())
