"""Soak of the overlapped front end: thousands of steps with three timesteps announced ahead (the bench's default depth).  orbf_step_end normally takes
every result word from the pinned buffer as soon as it carries the launch's sequence number (no end-of-kernel wait,
DESIGN section 7); MORB_POLL=0 makes it wait with hipStreamSynchronize instead.  Both must hand out exactly the same bytes
for every step -- a result word read too early, a stale frame or a recycled result set would show up as a different digest."""
import hashlib
import numpy as np
import pytest
from multi_orb_slam_amd import synth

pytestmark = pytest.mark.gpu
N_STEPS = 5000


def _run(monkeypatch, poll, isolated=False):
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, rt
    monkeypatch.setenv("MORB_POLL", "1" if poll else "0")       # read when the front end is created
    W, H, RING = 640, 480, 8
    fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=1000)] * 2, W, H)
    dev = []
    for t in range(RING):
        row = []
        for c in range(2):
            b = rt.DeviceBuffer(W * H); b.upload(synth.image(c, t, W, H)); row.append(b)
        dev.append(row)
    rt.device_sync()
    arg = lambda t: [(dev[t % RING][c].ptr, W) for c in range(2)]
    fe.copy_results = False                                       # results consumed in place, as the benchmark does
    digests = []
    AHEAD = 3
    if not isolated:
        for k in range(1, AHEAD):
            fe.announce(arg(k), resident=True)
    for t in range(N_STEPS):
        r = fe.step(arg(t), resident=True, next_images=None if isolated else arg(t + AHEAD))
        parts = [hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest()
                 for a in (r["match_of_feature"], r["kps"], r["desc"], r["uright"], r["cross"][0], r["cross"][1], r["cross"][2])]
        parts.append(repr((r["counts"], r["n_temporal"], r["n_cross"])).encode())
        digests.append(tuple(parts))
    fe.close()
    return digests


PARTS = ("match_of_feature", "kps", "desc", "uright", "cross_idx", "cross_best", "cross_second", "counts")


def _explain(a, b, bad):
    """which parts of which run left the pattern: the stream repeats every 8 frames, so a run's own digests of one lap earlier tell
    which of the two runs deviated"""
    out = []
    for t in bad[:10]:
        names = [PARTS[i] for i in range(len(PARTS)) if a[t][i] != b[t][i]]
        who = []
        if t >= 24:
            if a[t] != a[t - 8]: who.append("polled run deviates from its own lap before")
            if b[t] != b[t - 8]: who.append("synchronised run deviates from its own lap before")
        out.append("step %d: %s (%s)" % (t, ", ".join(names), "; ".join(who) or "?"))
    return "; ".join(out)


def test_polled_and_synchronised_result_pickup_agree_over_5000_steps(monkeypatch):
    a = _run(monkeypatch, True)
    b = _run(monkeypatch, False)
    bad = [t for t in range(N_STEPS) if a[t] != b[t]]
    assert not bad, "steps whose polled results differ from the synchronised run: %s" % _explain(a, b, bad)
    # the stream repeats every 8 frames: from the second lap on the digests repeat too (nothing leaks from step to step)
    assert all(a[t] == a[t - 8] for t in range(24, N_STEPS))
    assert len(set(a[16:24])) == 8


def test_isolated_steps_polled_and_synchronised_agree_over_5000_steps(monkeypatch):
    """No look-ahead: every step extracts its own images, and the camera-pair top-2 and the copy into the pinned result
    mirrors ride in the projection kernel's launch (k_project_side) -- their results are taken on the strength of the
    resolve's result words alone, so they must be complete by then, every time."""
    a = _run(monkeypatch, True, isolated=True)
    b = _run(monkeypatch, False, isolated=True)
    bad = [t for t in range(N_STEPS) if a[t] != b[t]]
    assert not bad, "isolated steps whose polled results differ from the synchronised run: %s" % _explain(a, b, bad)
    assert all(a[t] == a[t - 8] for t in range(24, N_STEPS))


def test_digest_of_a_stream_with_changing_motion_does_not_depend_on_the_ab_switches():
    """tools/soak_motion.py drives 600 overlapped steps whose motion (du, dv, th) changes every step and hashes every
    match list and cross-camera result.  The digest must be the same with the queries built on the device or by the host, the
    top-2 in FP4 or int8, the resolve monotone or Jacobi (the switches are read once per process: one process per setting)."""
    import os, subprocess, sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_motion.py")
    out = []
    for extra in ({}, {"MORB_MOTION_ON_DEVICE": "0", "MORB_TOP2_FP4": "0", "MORB_RESOLVE_MONO": "0"}):
        env = dict(os.environ); env.update(extra)
        r = subprocess.run([sys.executable, script, "600"], env=env, capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(r.stdout.strip().splitlines()[-1])
    assert out[0] == out[1] and "digest" in out[0], out
    assert int(out[0].split("temporal ")[1].split(",")[0]) > 600 * 300


def test_extraction_next_to_the_matrix_core_top2_on_one_hardware_queue():
    """The aggravated condition of profiles/r05/describe_defect.md as a regression test: ONE hardware queue
    (GPU_MAX_HW_QUEUES=1, where the build WITH the SLP vectorizer produced wrong descriptor bits in 65 % of the runs), four front ends on
    four host threads -- three extractor streams each, their describe / FAST / quadtree chains next to the rig-wide FP4 matrix-core top-2
    of the exchange on the matcher streams -- kept alive over hundreds of 10-step runs for 12 s: every keypoint record (position,
    response, angle: what k_fast_cells' candidates and the quadtree decide) and every descriptor of every step against the oracle.
    The product build (-fno-slp-vectorize, held by tests/test_isa_guard.py) must not show a single wrong row."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "describe_defect", "run_rig.py"), "--seconds", "12", "--persistent", "25", "--placement", "chain",
                        "--tag", "suite"], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["GPU_MAX_HW_QUEUES"] == "1" and out["lib"] == "libmorb.so"
    assert out["runs"] >= 50, out
    assert out["bad_runs"] == 0 and out["wrong_rows"] == 0 and not out["detail"], out
