"""orbf_step (include/orbf.h): the whole timestep as one native call must equal the oracle pipeline bit for bit."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libmorb: torch ships its own HIP runtime and has to be the one that initialises it,
              #  see the exchange test at the end of this file)
from multi_orb_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,nfs", [(320, 240, (300, 150)), (640, 480, (1000, 500)), (640, 480, (1000, 1000, 700)),
                                     (1280, 720, (2000, 2000)),    # = configs[2]
                                     (752, 480, (1200,)),          # one camera: no camera-pair matching at all
                                     (800, 250, (600, 600)),       # panorama-shaped levels: three root strips in the quadtree
                                     (640, 480, (2000, 50))])      # very unequal cameras
def test_native_step_equals_oracle_pipeline(w, h, nfs):
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, rt
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    params = [m.ExtractorParams(nfeatures=n) for n in nfs]
    fe = pipeline.FrontEnd(params, w, h)
    ofe = OracleFrontEnd(params, w, h)
    bufs = []
    for t in range(4):
        imgs = [synth.image(c, t, w, h) for c in range(len(nfs))]
        if t % 2 == 0:                                  # host images
            got = fe.step(imgs)
        else:                                           # HBM-resident images
            row = []
            for im in imgs:
                b = rt.DeviceBuffer(im.nbytes); b.upload(im); row.append(b)
            bufs.append(row)
            got = fe.step([(b.ptr, w) for b in row], resident=True)
        exp = ofe.step(imgs)
        assert_same_step(got, exp)
    assert got["n_temporal"] > 50 and sum(got["counts"]) > sum(nfs) // 2
    fe.close()


@pytest.mark.parametrize("w,h,nfs,resident,depth", [(320, 240, (300, 150), False, 1), (640, 480, (1000, 1000), True, 1),
                                                     (320, 240, (300, 150), True, 2), (640, 480, (1000, 1000), False, 2),
                                                     (640, 480, (1000, 1000), True, 3), (320, 240, (300, 150), False, 3)])
def test_overlapped_steps_equal_oracle_pipeline(w, h, nfs, resident, depth):
    """orbf_prefetch: the extraction of the next step (depth 1), the next two or the next three steps (on two / three extractor
    instances going round) runs next to this step's matching; results must not change."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, rt
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    params = [m.ExtractorParams(nfeatures=n) for n in nfs]
    fe = pipeline.FrontEnd(params, w, h)
    ofe = OracleFrontEnd(params, w, h)
    T = 8 if depth < 3 else 13      # (depth 3: long enough for extractors and result sets to go round more than once)
    frames = [[synth.image(c, t, w, h) for c in range(len(nfs))] for t in range(T)]
    dev = []
    if resident:
        for imgs in frames:
            row = []
            for im in imgs:
                b = rt.DeviceBuffer(im.nbytes); b.upload(im); row.append(b)
            dev.append(row)
    arg = (lambda t: [(b.ptr, w) for b in dev[t]]) if resident else (lambda t: frames[t])
    announced = 0                                     # index of the youngest step announced (or executed) so far
    for t in range(T):
        # keep `depth` future steps announced -- except step 4, which nobody announces (plain path in between)
        while announced < min(t + depth, T - 1) and announced + 1 != 4 and t != 4:
            announced += 1
            fe.announce(arg(announced), resident=resident)
        announced = max(announced, t)
        got = fe.step(arg(t), resident=resident)
        assert_same_step(got, ofe.step(frames[t]))
    assert got["n_temporal"] > 50
    fe.close()


@pytest.mark.parametrize("overlap,host_tree", [(False, False), (True, False), (False, True)])
def test_reference_calibration_undistorts_as_the_reference_frame_does(overlap, host_tree, monkeypatch):
    """OtherFiles/multi.yaml's calibration (k1 != 0): keypoints undistorted with cv::undistortPoints' iteration, image
    bounds from the undistorted corners, uRight from the undistorted x, depth read at the distorted pixel -- on the
    device, against the oracle's restatement of Frame::UndistortKeyPoints / ComputeImageBounds / ComputeStereoFromRGBD."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    from test_oracle_undistort import MULTI_YAML
    if host_tree:   # the synchronous path assembles the frame in the matcher's own kernel (k_frame_build_small) instead
        monkeypatch.setenv("MORB_HOST_OCTREE", "1")
    params = [m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=500)]       # Tracking.cc:144-145
    fe = pipeline.FrontEnd(params, 640, 480, calib=MULTI_YAML)
    ofe = OracleFrontEnd(params, 640, 480, calib=MULTI_YAML)
    frames = [[synth.image(c, t, 640, 480) for c in range(2)] for t in range(5)]
    for t in range(5):
        got = fe.step(frames[t], next_images=frames[t + 1] if overlap and t + 1 < 5 else None)
        assert_same_step(got, ofe.step(frames[t]))
    assert np.abs(got["un_x"] - got["kps"]["x"]).max() > 0.5 and got["n_temporal"] > 100
    fe.fe.set_calibration(None); ofe.calib = None                                       # and off again
    for t in range(2):
        got = fe.step(frames[t]); exp = ofe.step(frames[t])
        if t == 1:
            assert_same_step(got, exp)
    assert np.array_equal(got["un_x"], got["kps"]["x"])
    fe.close()


@pytest.mark.parametrize("depth", [0, 2])
def test_six_camera_rig_takes_the_multi_kernel_frame_path(depth):
    """More than 4 cameras: the frame is assembled by the matcher's own kernels from the extractor's per-camera outputs
    (camera table finished on the device from the counts in HBM), also with two timesteps announced ahead."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    nfs = (300, 150, 200, 300, 100, 250)
    params = [m.ExtractorParams(nfeatures=n) for n in nfs]
    fe = pipeline.FrontEnd(params, 320, 240)
    ofe = OracleFrontEnd(params, 320, 240)
    T = 6
    frames = [[synth.image(c, t, 320, 240) for c in range(len(nfs))] for t in range(T)]
    announced = 0
    for t in range(T):
        while announced < min(t + depth, T - 1):
            announced += 1
            fe.announce(frames[announced])
        announced = max(announced, t)
        assert_same_step(fe.step(frames[t]), ofe.step(frames[t]))
    fe.close()


@pytest.mark.parametrize("kind", list(synth.FAMILIES))
def test_whole_steps_on_the_image_families(kind):
    """The image families of round 5 (synth.family_image: 1/f-like noise, dithered ramps, soft edges, saturated regions, contrast
    at the FAST thresholds) through the WHOLE step -- extraction, frame assembly with stereo, the temporal search with its rotation
    histogram, the camera-pair top-2 --, two cameras at 640x480, isolated steps and steps announced ahead, against the oracle."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    w, h, T = 640, 480, 5
    params = [m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=500)]
    frames = [[synth.family_image(kind, c, t, w, h) for c in range(2)] for t in range(T)]
    for depth in (0, 2):
        fe = pipeline.FrontEnd(params, w, h); ofe = OracleFrontEnd(params, w, h)
        announced = 0
        for t in range(T):
            while announced < min(t + depth, T - 1):
                announced += 1
                fe.announce(frames[announced])
            announced = max(announced, t)
            got = fe.step(frames[t])
            assert_same_step(got, ofe.step(frames[t]))
        assert sum(got["counts"]) > 900, (kind, got["counts"])
        fe.close()


def test_overlap_survives_the_host_quadtree_fallback():
    """Noise frames put more candidates on level 0 than the device quadtree takes (its limit lowered to 4096 here; 65535 in
    the product): the kernel reports 'outside my limits' and the step is redone on the host path -- with the next step's
    extraction already in flight its images have to be uploaded again."""
    import os
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    w, h = 640, 480
    params = [m.ExtractorParams(nfeatures=500)] * 2

    def noise(c, t):
        r = synth.hash32(np.arange(w * h, dtype=np.uint64) + np.uint64(1000 * t + 17 * c))
        return (r % 256).astype(np.uint8).reshape(h, w)

    frames = [[noise(c, t) for c in range(2)] for t in range(2)] + [[synth.image(c, t, w, h) for c in range(2)] for t in range(2, 5)]
    os.environ["MORB_OCT_MAX_KEYS"] = "4096"
    try:
        fe = pipeline.FrontEnd(params, w, h)
    finally:
        os.environ.pop("MORB_OCT_MAX_KEYS")
    ofe = OracleFrontEnd(params, w, h)
    for t in range(5):
        got = fe.step(frames[t], next_images=frames[t + 1] if t + 1 < 5 else None)
        assert_same_step(got, ofe.step(frames[t]))
        if t == 0:
            assert len(fe.ex.debug_candidates(0, 0)) > 4096      # the case this test is about
    fe.close()


def test_prefetch_of_other_images_is_dropped():
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
    fe = pipeline.FrontEnd(params, 320, 240)
    ofe = OracleFrontEnd(params, 320, 240)
    frames = [[synth.image(c, t, 320, 240) for c in range(2)] for t in range(4)]
    assert_same_step(fe.step(frames[0], next_images=frames[3]), ofe.step(frames[0]))   # announces frame 3 ...
    assert_same_step(fe.step(frames[1], next_images=frames[2]), ofe.step(frames[1]))   # ... but frame 1 comes
    assert_same_step(fe.step(frames[2]), ofe.step(frames[2]))
    fe.reset(); ofe = OracleFrontEnd(params, 320, 240)                                   # reset drops what is in flight
    assert_same_step(fe.step(frames[3], next_images=frames[0]), ofe.step(frames[3]))
    fe.reset(); ofe = OracleFrontEnd(params, 320, 240)
    assert_same_step(fe.step(frames[1]), ofe.step(frames[1]))
    fe.close()


def test_native_step_with_host_resolve_keeps_cross_results(monkeypatch):
    """The exact host resolve (fallback of the device resolve) re-projects through the matcher's staging buffers; the
    camera-pair top-2 of the same step must survive that."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    monkeypatch.setenv("MORB_HOST_RESOLVE", "1")
    params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
    fe = pipeline.FrontEnd(params, 320, 240)
    ofe = OracleFrontEnd(params, 320, 240)
    for t in range(3):
        imgs = [synth.image(c, t, 320, 240) for c in range(2)]
        assert_same_step(fe.step(imgs), ofe.step(imgs))
    fe.close()


def test_native_step_with_an_empty_camera():
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd.frontend import NativeFrontEnd
    import oracle
    fe = NativeFrontEnd([m.ExtractorParams(nfeatures=300)] * 2, 320, 240)
    img = synth.image(0, 0, 320, 240)
    r = fe.step([img, None])
    assert r["counts"][1] == 0 and r["counts"][0] > 100
    ok, od = oracle.extract(img, nfeatures=300)
    assert r["kps"].tobytes() == ok.tobytes() and np.array_equal(r["desc"], od)
    assert (r["uright"] == -1).all()                    # no depth image set
    bi, bd, sd = r["cross"]
    assert (bi == -1).all() and (bd == 256).all()        # nothing to match against
    fe.close()


def test_full_size_rig_properties():
    """configs[4]: 8 synthetic 1920x1080 streams, 4000 features per camera, through the native front end (the > 4-camera
    path: dense levels of up to 9400 candidates in the device quadtree, multi-kernel frame assembly, the 32 000-feature
    first-come search with its claim tables in HBM, the 32k x 28k camera-pair top-2 on the matrix cores).  The WHOLE rig is
    pinned against the oracle (one host thread per camera; a few seconds): keypoints, descriptors, stereo, the temporal
    search and the complete cross-camera top-2 -- then batch independence and determinism."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    W, H, NF, NC = 1920, 1080, 4000, 8
    params = [m.ExtractorParams(nfeatures=NF)] * NC
    frames = [[synth.image(c, t, W, H) for c in range(NC)] for t in range(2)]

    def run():
        fe = pipeline.FrontEnd(params, W, H)
        out = [fe.step(frames[t]) for t in range(2)]
        fe.close()
        return out

    a = run()
    ofe = OracleFrontEnd(params, W, H, cam_threads=True)
    for t in range(2):
        assert_same_step(a[t], ofe.step(frames[t]))
    ofe.pool.shutdown()
    r1 = a[1]
    counts = r1["counts"]
    assert len(counts) == NC and all(NF * 0.95 < c <= NF + 32 for c in counts)
    assert r1["n_temporal"] > 0.5 * sum(counts) and len(r1["cross"][0]) == sum(counts)
    off = np.concatenate([[0], np.cumsum(counts)])
    # batch independence: a single-camera extractor gives the same
    ex = m.Extractor([params[0]], W, H)
    k6, d6 = ex.extract([frames[1][6]])[0]
    ex.close()
    assert r1["kps"][off[6]:off[7]].tobytes() == k6.tobytes() and np.array_equal(r1["desc"][off[6]:off[7]], d6)
    # determinism: a second front end reproduces every byte
    b = run()
    for x, y in zip(a, b):
        assert x["kps"].tobytes() == y["kps"].tobytes() and np.array_equal(x["desc"], y["desc"])
        assert np.array_equal(x["match_of_feature"], y["match_of_feature"]) and all(np.array_equal(u, v) for u, v in zip(x["cross"], y["cross"]))


def test_single_rank_rccl_exchange_equals_oracle():
    """The multi-GPU step on ONE rank (world-size-1 RCCL group): descriptors leave through orbf_export_block, come back
    through one all-gather and are matched by orbm_cross_top2_gathered -- late (after the step) while nothing runs ahead,
    early (enqueued between orbf_step_begin and orbf_step_end) once the extraction of announced steps does."""
    import os
    import torch
    import torch.distributed as dist
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from multi_orb_slam_amd.dist import DescriptorExchange
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    import datetime, socket
    with socket.socket() as sk:      # a port nobody holds right now (the GPU boxes are shared machines)
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    try:
        torch.cuda.set_device(0)
        params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
        fe = pipeline.FrontEnd(params, 320, 240)
        fe.gather = DescriptorExchange(torch.device("cuda", 0), dist); fe.world = 2   # forced exchange on one rank
        ofe = OracleFrontEnd(params, 320, 240)
        T = 7
        frames = [[synth.image(c, t, 320, 240) for c in range(2)] for t in range(T)]
        fe.announce(frames[1])
        for t in range(T):
            if t >= 3:      # a busy default stream delays the collective: the gathered matching must wait for it on the device
                torch.cuda._sleep(4_000_000)
            got = fe.step(frames[t], next_images=frames[t + 2] if t + 2 < T else None)
            assert_same_step(got, ofe.step(frames[t]))
        assert 0 < fe.early_exchanges < T          # both orders of the exchange were exercised
        # the same steps with the all-gather issued natively from inside the step (RCCL C API, communicator of its own)
        assert fe.enable_native_exchange(dist, torch.device("cuda", 0)) and fe.fe.exchange_world == 1
        fe.reset(); ofe = OracleFrontEnd(params, 320, 240)
        fe.announce(frames[1])
        for t in range(T):
            got = fe.step(frames[t], next_images=frames[t + 2] if t + 2 < T else None)
            assert got["rig_counts"] == got["counts"]
            assert_same_step(got, ofe.step(frames[t]))
        fe.fe.exchange_shutdown(); fe.native_exchange = False
        assert fe.fe.exchange_world == 0
        fe.close()
        # A block that went out with its extraction chain BEFORE the device quadtree's fallback was known carries the mark and is
        # shipped again at the end of its step (frontend.hip: ORBM_BLOCK_REDO): noise in steps 1 and 2 (two redone steps in a row)
        # with two steps announced ahead, in both placements of the exchange, over the real transport.
        w, h = 640, 480
        params2 = [m.ExtractorParams(nfeatures=500)] * 2

        def noise(c, t):
            r = synth.hash32(np.arange(w * h, dtype=np.uint64) + np.uint64(1000 * t + 17 * c))
            return (r % 256).astype(np.uint8).reshape(h, w)

        frames2 = [[noise(c, t) if t in (1, 2) else synth.image(c, t, w, h) for c in range(2)] for t in range(6)]
        for placement in ("chain", "inline"):
            os.environ["MORB_OCT_MAX_KEYS"] = "4096"; os.environ["MORB_EXCHANGE_PLACEMENT"] = placement
            try:
                fe2 = pipeline.FrontEnd(params2, w, h)
                assert fe2.enable_native_exchange(dist, torch.device("cuda", 0))
            finally:
                os.environ.pop("MORB_OCT_MAX_KEYS"); os.environ.pop("MORB_EXCHANGE_PLACEMENT")
            assert fe2.fe.exchange_placement == (3 if placement == "chain" else 1)
            ofe2 = OracleFrontEnd(params2, w, h)
            fe2.announce(frames2[1])
            for t in range(6):
                got = fe2.step(frames2[t], next_images=frames2[t + 2] if t + 2 < 6 else None)
                assert got["rig_counts"] == got["counts"], (placement, t)
                assert_same_step(got, ofe2.step(frames2[t]))
            assert fe2.fe.debug_exchange_redos() == (2 if placement == "chain" else 0)   # (behind the search a block only goes out when it is final)
            # with an exchange an announcement is binding: the block of the announced images has been shipped
            if placement == "chain":
                fe2.announce(frames2[0])
                fe2.step(frames2[3])              # (extracts the announced frame 0 as the NEXT step and ships its block)
                ofe2.step(frames2[3])
                with pytest.raises(m.OrbError, match="announcements are binding"):
                    fe2.step(frames2[4])
                # the refused call left no step behind (its number went back): the step the ranks agreed on can still be made
                assert_same_step(fe2.step(frames2[0]), ofe2.step(frames2[0]))
            fe2.fe.exchange_shutdown(); fe2.native_exchange = False
            fe2.close()
        # Communicators: the steps in flight and the re-shipped blocks need INDEPENDENT communicators.  Without ncclCommSplit
        # (MORB_EXCHANGE_ONE_COMM=split-off) they are made with fresh ids gathered over the first one -- placement 3 as usual; when none can
        # be made (=1) every rank falls back to placement 1 over the one communicator, where a block only ships when it is final: the
        # same noise steps then need no re-shipment at all.
        for one, want_placement, want_redos in (("split-off", 3, 2), ("1", 1, 0)):
            os.environ["MORB_OCT_MAX_KEYS"] = "4096"; os.environ["MORB_EXCHANGE_ONE_COMM"] = one
            try:
                fe3 = pipeline.FrontEnd(params2, w, h)
                assert fe3.enable_native_exchange(dist, torch.device("cuda", 0))
            finally:
                os.environ.pop("MORB_OCT_MAX_KEYS"); os.environ.pop("MORB_EXCHANGE_ONE_COMM")
            assert fe3.fe.exchange_placement == want_placement, one
            ofe3 = OracleFrontEnd(params2, w, h)
            fe3.announce(frames2[1])
            for t in range(6):
                got = fe3.step(frames2[t], next_images=frames2[t + 2] if t + 2 < 6 else None)
                assert got["rig_counts"] == got["counts"], (one, t)
                assert_same_step(got, ofe3.step(frames2[t]))
            assert fe3.fe.debug_exchange_redos() == want_redos, one
            fe3.fe.exchange_shutdown(); fe3.native_exchange = False
            fe3.close()
    finally:
        dist.destroy_process_group()


def test_results_consumed_in_place_are_the_same_arrays():
    """copy_results = False (the timed loop of bench.py): the step hands out lazily built views of the native pinned buffers."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
    fe = pipeline.FrontEnd(params, 320, 240); ofe = OracleFrontEnd(params, 320, 240)
    fe.copy_results = False
    for t in range(3):
        imgs = [synth.image(c, t, 320, 240) for c in range(2)]
        got = fe.step(imgs)
        assert "kps" in got and "cross" in got and got.get("nothing") is None and got["n_total"] == sum(got["counts"])
        exp = ofe.step(imgs)
        assert got["n_cross"] >= 0
        assert_same_step(got.materialise(), exp)
    # a field nobody asked for before the front end's next step is refused afterwards (its record and buffers have been reused);
    # what was asked for in time stays readable
    first = fe.step([synth.image(c, 3, 320, 240) for c in range(2)])
    n_first = first["n_total"]; counts_first = first["counts"]
    fe.step([synth.image(c, 4, 320, 240) for c in range(2)])
    assert first["n_total"] == n_first and first["counts"] == counts_first
    with pytest.raises(RuntimeError):
        first["kps"]
    fe.close()


@pytest.mark.parametrize("world,n_cams,w,h,nf,ahead", [(4, 4, 640, 480, 1000, 0), (4, 4, 640, 480, 1000, 2), (2, 4, 640, 480, 1000, 1),
                                                       (4, 8, 320, 240, 300, 0)])
def test_rig_sharded_over_ranks_through_the_gathered_blocks(world, n_cams, w, h, nf, ahead):
    """configs[3]: a 4-camera 640x480 rig at 1000 features per camera, one camera per rank.  The ranks are `world` front ends
    on this one device; what RCCL's all-gather would do is done by device-to-device copies of every front end's REAL export
    block (descriptor rows of the step's frame with whatever earlier steps left behind the counted rows, count trailer) into
    one receive buffer, and every "rank" then runs orbm_cross_top2_gathered on it.  Keypoints, descriptors, stereo, temporal
    matches and the rig-wide cross-camera top-2 of every rank must equal the oracle's.  The reference analogue of the shard:
    the per-camera extractor calls of src/Frame.cc:182-185."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, rt
    from multi_orb_slam_amd.dist import shard_cameras, BLOCK_TRAILER
    from multi_orb_slam_amd.frontend import SKIP_CROSS
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    per = n_cams // world
    fes, ofes = [], []
    for r in range(world):
        mine = shard_cameras(n_cams, world, r)
        assert len(mine) == per
        params = [m.ExtractorParams(nfeatures=nf)] * per
        fes.append(pipeline.FrontEnd(params, w, h, rank=r, world_size=world, global_cams=mine))
        ofes.append(OracleFrontEnd(params, w, h, mine))
    T = 4
    frames = [{g: synth.image(g, t, w, h) for g in range(n_cams)} for t in range(T)]
    mo = (pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ)
    recv = None
    for t in range(T):
        got = []
        for r, fe in enumerate(fes):
            for a in range(t + 1, min(t + ahead, T - 1) + 1):       # keep `ahead` future steps announced (a FIFO)
                if a > getattr(fe, "_announced", 0):
                    fe.announce([frames[a][g] for g in fe.global_cams]); fe._announced = a
            res = fe.fe.step([frames[t][g] for g in fe.global_cams], None, SKIP_CROSS, motion=mo)
            p, nbytes, rows = fe.fe.export_block()
            assert nbytes == rows * 32 + BLOCK_TRAILER and rows >= per * nf
            if recv is None:
                recv = rt.DeviceBuffer(world * nbytes)
                recv.upload(np.full(world * nbytes, 0x5A, np.uint8))
            rt.memcpy_d2d(recv.ptr + r * nbytes, p, nbytes)           # this rank's slice of the "all-gather"
            got.append(res)
        rt.device_sync()
        desc_of = {}
        for r, fe in enumerate(fes):
            off = np.concatenate([[0], np.cumsum(got[r]["counts"])])
            for c, g in enumerate(fe.global_cams):
                desc_of[g] = got[r]["desc"][off[c]:off[c + 1]]
        for r, fe in enumerate(fes):
            bi, bd, sd, cnts = fe.mt.cross_top2_gathered(recv.ptr, world, nbytes, rows, per, r)
            assert cnts == [len(desc_of[g]) for g in range(n_cams)]
            got[r]["cross"] = (bi, bd, sd)
            got[r]["n_cross"] = int(pipeline.accept_cross(bd, sd).sum())
            exp = ofes[r].step([frames[t][g] for g in fe.global_cams],
                               other_descs=lambda c, fe=fe: [desc_of[g] for g in range(n_cams) if g != fe.global_cams[c]])
            assert_same_step(got[r], exp)
            assert len(bi) == sum(got[r]["counts"]) and (bi >= 0).all()
    assert all(g["n_temporal"] > 50 for g in got) and sum(g["n_cross"] for g in got) >= 0
    for fe in fes:
        fe.close()


def test_refilled_host_buffer_is_extracted_again_not_served_stale():
    """orbf_prefetch identifies an announced frame by its pointers -- and, for host images, by a content fingerprint taken
    when the upload was enqueued: a caller that announces a buffer and then REFILLS it before the step arrives gets the new
    content extracted, not the extraction that ran ahead on the old bytes."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
    fe = pipeline.FrontEnd(params, 320, 240)
    ofe = OracleFrontEnd(params, 320, 240)
    frames = [[synth.image(c, t, 320, 240) for c in range(2)] for t in range(5)]
    ring = [[frames[t][c].copy() for c in range(2)] for t in range(3)]      # three reusable host buffers per camera
    assert_same_step(fe.step(ring[0], next_images=ring[1]), ofe.step(frames[0]))       # step 1's extraction now runs ahead ...
    for c in range(2):
        ring[1][c][:] = frames[3][c]                                                   # ... and the caller overwrites that buffer
    assert_same_step(fe.step(ring[1]), ofe.step(frames[3]))                            # the step must see frame 3
    assert_same_step(fe.step(ring[2], next_images=ring[0]), ofe.step(frames[2]))       # unchanged buffers still ride the prefetch
    assert_same_step(fe.step(ring[0]), ofe.step(frames[0]))
    fe.close()


def test_recycled_device_buffer_with_a_generation_is_extracted_again():
    """A device image cannot be fingerprinted from the host: a caller that recycles HBM buffers says so with
    orbf_image::generation (the frame's sequence number).  The extraction that ran ahead on a buffer's OLD content is then
    not served to the step that arrives with the buffer refilled under a new generation; the same (pointer, generation)
    still rides the prefetch."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, rt
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    W, H = 320, 240
    params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
    fe = pipeline.FrontEnd(params, W, H)
    ofe = OracleFrontEnd(params, W, H)
    frames = [[synth.image(c, t, W, H) for c in range(2)] for t in range(5)]
    ring = [[rt.DeviceBuffer(W * H) for c in range(2)] for t in range(2)]      # two reusable HBM buffers per camera
    def fill(slot, t):
        for c in range(2):
            ring[slot][c].upload(frames[t][c])
    def args(slot, gen):
        return [(ring[slot][c].ptr, W, gen) for c in range(2)]
    fill(0, 0); fill(1, 1)
    assert_same_step(fe.step(args(0, 1), resident=True, next_images=args(1, 2)), ofe.step(frames[0]))   # frame 1 runs ahead
    assert_same_step(fe.step(args(1, 2), resident=True, next_images=args(0, 1)), ofe.step(frames[1]))   # served from the prefetch; slot 0 (frame 0) announced again
    rt.device_sync()
    fill(0, 3)                                                         # ... the caller refills slot 0 with frame 3, generation 4
    assert_same_step(fe.step(args(0, 4), resident=True), ofe.step(frames[3]))                             # must see frame 3
    fill(1, 4)
    assert_same_step(fe.step(args(1, 5), resident=True), ofe.step(frames[4]))
    fe.close()


def _loopback_rig(world, n_cams, w, h, nf, ahead, T=6, group=None, delayed_rank=None, delay_s=0.0, probe=False, frame_fn=None):
    """`world` front ends on this one device, one host thread each, the cameras of ONE rig sharded over them, exchanging their
    export blocks from inside the native step through the in-process loopback transport.  Optionally one rank sleeps `delay_s`
    before every step (a slow peer) and every rank records when its search / its exchange had finished on the device
    (orbf_debug_exchange_timing).  Every rank's every step is then held against the oracle.
    -> (results[r][t], timings[r][t] = (search_us, exchange_us) or None, placements[r])"""
    import threading
    import time
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from multi_orb_slam_amd.dist import shard_cameras
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    per = n_cams // world
    frames = [{g: (frame_fn(g, t) if frame_fn else synth.image(g, t, w, h)) for g in range(n_cams)} for t in range(T)]
    results = [[None] * T for _ in range(world)]
    timings = [[None] * T for _ in range(world)]
    placements = [None] * world
    errors = []
    group = group if group is not None else 1000 + world * 10 + ahead

    def rank_main(r):
        try:
            mine = shard_cameras(n_cams, world, r)
            fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=nf)] * per, w, h, rank=r, world_size=world, global_cams=mine)
            fe.fe.exchange_init_loopback(group, world, r)
            fe.native_exchange = True
            assert fe.fe.exchange_world == world
            placements[r] = fe.fe.exchange_placement
            if probe:
                fe.fe.debug_exchange_timing(True)
            announced = 0
            for t in range(T):
                if r == delayed_rank:
                    time.sleep(delay_s)       # (releases the interpreter lock: the other ranks' threads keep stepping)
                while announced < min(t + ahead, T - 1):
                    announced += 1
                    fe.announce([frames[announced][g] for g in mine])
                announced = max(announced, t)
                results[r][t] = fe.step([frames[t][g] for g in mine])
                if probe:
                    timings[r][t] = fe.fe.debug_exchange_us()
            fe.fe.exchange_shutdown()
            fe.close()
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errors and not any(th.is_alive() for th in threads), errors
    ofes = [OracleFrontEnd([m.ExtractorParams(nfeatures=nf)] * per, w, h, shard_cameras(n_cams, world, r)) for r in range(world)]
    for t in range(T):
        desc_of = {}
        for r in range(world):
            got = results[r][t]
            off = np.concatenate([[0], np.cumsum(got["counts"])])
            for c, g in enumerate(shard_cameras(n_cams, world, r)):
                desc_of[g] = got["desc"][off[c]:off[c + 1]]
        for r in range(world):
            mine = shard_cameras(n_cams, world, r)
            got = results[r][t]
            assert got["rig_counts"] == [len(desc_of[g]) for g in range(n_cams)]
            exp = ofes[r].step([frames[t][g] for g in mine],
                               other_descs=lambda c, mine=mine: [desc_of[g] for g in range(n_cams) if g != mine[c]])
            assert_same_step(got, exp)
    assert all(results[r][T - 1]["n_temporal"] > 30 for r in range(world))
    return results, timings, placements


@pytest.mark.parametrize("world,n_cams,w,h,nf,ahead", [(4, 4, 640, 480, 1000, 0), (4, 4, 640, 480, 1000, 2), (2, 4, 320, 240, 300, 2), (8, 8, 320, 240, 200, 1),
                                                       (2, 10, 960, 540, 1800, 1)])   # (five cameras per rank: the large-rig frame assembly and pyramid behind an exchange)
def test_world_size_n_native_steps_over_the_loopback_exchange(world, n_cams, w, h, nf, ahead):
    """The multi-GPU step with world > 1, end to end through orbf_step: `world` front ends (one host thread each, the cameras
    of ONE rig sharded over them -- configs[3]: one 640x480 camera @1000 per rank) exchange their export blocks from INSIDE the
    native step, exactly as over RCCL, through the in-process loopback transport (RCCL refuses two ranks on one GPU).  Every
    rank's keypoints, descriptors, stereo, temporal matches and rig-wide cross-camera top-2 must equal the oracle's; with
    steps announced ahead some ranks ship their block early (between begin and end), others late -- any mix must work."""
    _results, _timings, placements = _loopback_rig(world, n_cams, w, h, nf, ahead)
    assert placements == [3] * world          # the default at every world size: at the tail of the step's extraction chain


@pytest.mark.parametrize("placement", ["chain", "inline"])
def test_a_block_shipped_before_its_extraction_fell_back_is_shipped_again(placement, monkeypatch):
    """A step's exchange goes out with its extraction chain, before anybody knows whether the device quadtree stayed inside its
    limits.  Camera 2 (rank 2 of four) sees noise in step 1 -- more candidates on level 0 than the quadtree takes (limit lowered
    to 4096 here) --, its step is redone on the host path, the block it had shipped carries the mark and EVERY rank ships its
    final block a second time at the end of that step: all ranks' rig-wide top-2 must equal the oracle's on the final
    descriptors, in the step with the redo and around it, with two steps announced ahead.  (Behind the search the same with two
    redone steps in a row; with the chain the LOOPBACK transport cannot do that -- its rendezvous blocks the host, a rank that
    stops extracting ahead after a fallback issues its later exchanges later than its peers, and a second redo would then wait
    for ranks that wait for it.  RCCL calls return at once: test_single_rank_rccl_exchange_equals_oracle runs that case.)"""
    monkeypatch.setenv("MORB_EXCHANGE_PLACEMENT", placement)
    monkeypatch.setenv("MORB_OCT_MAX_KEYS", "4096")
    w, h = 640, 480

    def frame(g, t):
        if g == 2 and t in ((1, 2) if placement == "inline" else (1,)):
            r = synth.hash32(np.arange(w * h, dtype=np.uint64) + np.uint64(1000 * t + 17 * g))
            return (r % 256).astype(np.uint8).reshape(h, w)
        return synth.image(g, t, w, h)

    results, _t, placements = _loopback_rig(4, 4, w, h, 500, ahead=2, T=6, group=2100 + (placement == "inline"), frame_fn=frame)
    assert placements == [1 if placement == "inline" else 3] * 4
    assert results[2][1]["counts"][0] > 400 and results[0][1]["rig_counts"][2] == results[2][1]["counts"][0]


@pytest.mark.parametrize("placement", ["chain", "inline"])
def test_slow_peer_delays_the_end_of_the_step_never_the_local_search(placement, monkeypatch):
    """configs[3]'s rig over four ranks with one rank arriving 1 ms late at EVERY step, in both arrangements of the exchange: at the
    tail of the step's extraction chain (the default: issued two steps ahead of the step's matching) and behind the step's search
    on the matcher's stream (rounds 2-4).  In either the local search never waits for the late peer; behind the search the end of
    the step carries the peer's delay.  Results stay bit-identical with the oracle on every rank and step."""
    monkeypatch.setenv("MORB_EXCHANGE_PLACEMENT", placement)
    world, delay = 4, 1.0e-3
    _results, timings, placements = _loopback_rig(world, 4, 640, 480, 1000, ahead=2, T=10, group=1900 + (placement == "inline"),
                                                  delayed_rank=1, delay_s=delay, probe=True)
    assert placements == [1 if placement == "inline" else 3] * world
    for r in (0, 2, 3):
        search = np.median([timings[r][t][0] for t in range(3, 10)])
        exch = np.median([timings[r][t][1] for t in range(3, 10)])
        assert 0 < search < 0.5 * delay * 1e6, (r, search, exch)          # the local search never waited for the late peer
        if placement == "inline":
            assert exch - search > 0.4 * delay * 1e6, (r, search, exch)   # ... the end of the step did
        # (with the chain a step's exchange is issued two steps earlier: what that absorbs is a peer's JITTER of up to the look-ahead;
        # a peer that is late at every step sets the pace in any arrangement -- its block for step t cannot exist before it has
        # extracted step t -- so nothing is asserted about `exch` here; the figures are printed)
    print("slow peer (%s): punctual ranks' median search done %.0f us, exchange done %.0f us after the start of the step's matching"
          % (placement, np.median([timings[r][t][0] for r in (0, 2, 3) for t in range(3, 10)]),
             np.median([timings[r][t][1] for r in (0, 2, 3) for t in range(3, 10)])))


def test_level0_read_in_place_on_the_resize_chain_pyramid():
    """Large rigs read pyramid level 0 in the caller's device buffers instead of copying it (round 4, extractor.hip: k_set_l0).  The
    pyramid arrangement is chosen once per process, so tests/inplace_leg.py runs as a child with the large-rig tile launches forced at
    640x480: tight and padded pitches, host images in between, a misaligned buffer and an odd pitch (copied as before), steps
    announced ahead -- every step equal to the oracle, and the inspection hook says which runs really read in place."""
    import os
    import subprocess
    import sys
    leg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inplace_leg.py")
    env = dict(os.environ, MORB_PYR_CHAIN="1")   # (the two tile launches of large rigs, as on 8 x 1080p)
    out = subprocess.run([sys.executable, leg], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "inplace_leg ok: level 0 read in place in 4 of 6 isolated steps on device images" in out.stdout, out.stdout + out.stderr


def test_native_stream_loop_equals_the_step_by_step_run():
    """orbf_run_stream (the synthetic-stream loop in one native call, what bench.py times as `value_c_abi_loop`) must do exactly
    what the binding's loop over announce + step does: same per-step feature, temporal-match and accepted cross-match counts,
    with two steps announced ahead and with none -- and the overlapped run must equal the isolated one step for step."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, rt
    from multi_orb_slam_amd.matcher import TH_LOW
    W, H, RING, N = 640, 480, 8, 40
    params = [m.ExtractorParams(nfeatures=1000)] * 2
    dev = []
    for t in range(RING):
        row = []
        for c in range(2):
            b = rt.DeviceBuffer(W * H); b.upload(synth.image(c, t, W, H)); row.append(b)
        dev.append(row)
    ring = [[(dev[t][c].ptr, W, H, W, 1) for c in range(2)] for t in range(RING)]
    motion = (pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ)

    def mix(h, n, nt, nx):
        h ^= ((n & 0xffffffff) << 40) ^ ((nt & 0xffffffff) << 20) ^ (nx & 0xffffffff)
        h = (h * 0x9E3779B97F4A7C15) & 0xffffffffffffffff
        return h ^ (h >> 29)

    # step by step through the binding
    fe = pipeline.FrontEnd(params, W, H)
    fe.copy_results = False
    arg = lambda t: [(dev[t % RING][c].ptr, W) for c in range(2)]
    fe.announce(arg(1), resident=True)
    tot = [0, 0, 0]; dig = 0
    for t in range(N):
        r = fe.step(arg(t), resident=True, next_images=arg(t + 2))
        tot[0] += r["n_total"]; tot[1] += r["n_temporal"]; tot[2] += r["n_cross"]
        dig = mix(dig, r["n_total"], r["n_temporal"], r["n_cross"])
    fe.close()
    for ahead in (2, 0):
        fe = pipeline.FrontEnd(params, W, H)
        upto = -1
        acc = [0, 0, 0]; digests = []
        for t0 in (0, 13, 27):     # three consecutive calls continue one stream
            n = {0: 13, 13: 14, 27: N - 27}[t0]
            st, upto = fe.fe.run_stream(ring, t0, n, ahead, upto, motion, TH_LOW, pipeline.BOW_RATIO)
            acc[0] += st["features"]; acc[1] += st["temporal_matches"]; acc[2] += st["cross_accepted"]
            digests.append(st["digest"])
            assert st["seconds"] > 0
        fe.close()
        assert acc == tot, (ahead, acc, tot)
    assert tot[1] > 30 * 1000 and tot[2] > 0
    # one call over the whole stream reproduces the order-sensitive digest of the step-by-step run
    fe = pipeline.FrontEnd(params, W, H)
    st, _ = fe.fe.run_stream(ring, 0, N, 2, -1, motion, TH_LOW, pipeline.BOW_RATIO)
    fe.close()
    assert st["digest"] == dig


@pytest.mark.parametrize("handles", [1, 2])
def test_fewer_extractor_instances_take_fewer_steps_ahead(handles, monkeypatch):
    """MORB_AHEAD_DEPTH / orbf_create_depth: a front end with one or two extractor instances accepts that many timesteps ahead
    (orbf_ahead_depth), refuses one more, and returns the oracle's results at its full depth."""
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from multi_orb_slam_amd._lib import OrbError
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    monkeypatch.setenv("MORB_AHEAD_DEPTH", str(handles))
    w, h = 320, 240
    params = [m.ExtractorParams(nfeatures=300), m.ExtractorParams(nfeatures=150)]
    fe = pipeline.FrontEnd(params, w, h)
    ofe = OracleFrontEnd(params, w, h)
    assert fe.fe.ahead_depth == handles
    T = 7
    frames = [[synth.image(c, t, w, h) for c in range(2)] for t in range(T)]
    announced = 0
    for t in range(T):
        while announced < min(t + handles, T - 1):
            announced += 1
            fe.announce(frames[announced])
        got = fe.step(frames[t])
        assert_same_step(got, ofe.step(frames[t]))
    # one announcement too many is refused (the FIFO holds `handles` steps beyond the next one)
    fe2 = pipeline.FrontEnd(params, w, h)
    for k in range(handles + 1):
        fe2.announce(frames[k + 1])
    with pytest.raises(OrbError):
        fe2.announce(frames[handles + 2])
    fe.close(); fe2.close()


@pytest.mark.parametrize("ahead", [0, 2])
def test_motion_queries_built_on_the_device_are_the_hosts_records(ahead):
    """orbf_step_motion lets the projection kernel build its queries from the previous frame in HBM.  The records the step hands back
    (orbf_result::queries, written by the host while it waits) must be orbm_queries_from_motion of the previous step's results, and
    with ORBF_NO_QUERY_RECORDS the pointer stays NULL while the matches stay the same.  (That the matches are the oracle's search over
    exactly those records is what every oracle-pipeline test of this file checks: the oracle leg builds its queries with
    make_queries.)"""
    import ctypes as C
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from multi_orb_slam_amd.extractor import tables
    from multi_orb_slam_amd.frontend import NativeFrontEnd, NO_QUERY_RECORDS
    from multi_orb_slam_amd.matcher import QUERY_DTYPE
    W, H, T = 640, 480, 6
    params = [m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=600)]
    frames = [[synth.image(c, t, W, H) for c in range(2)] for t in range(T)]
    motion = (pipeline.MOTION[0], pipeline.MOTION[1], pipeline.TH_PROJ)
    scale = np.asarray(tables(params[0])["scale"], np.float32)
    runs = {}
    for flags in (0, NO_QUERY_RECORDS):
        fe = NativeFrontEnd(params, W, H)
        fe.configure(pipeline.MBF, 100, True)
        out = []
        announced = 0
        for t in range(T):
            while announced < min(t + ahead, T - 1):
                announced += 1
                fe.prefetch(frames[announced])
            r = fe.step(frames[t], flags=flags, motion=motion)
            qp = fe._res.queries
            recs = None
            if qp:
                nb = r["n_queries"] * QUERY_DTYPE.itemsize
                recs = np.ctypeslib.as_array(C.cast(qp, C.POINTER(C.c_uint8)), shape=(nb,)).view(QUERY_DTYPE).copy()
            out.append((r, recs))
        runs[flags] = out
        fe.close()
    for t in range(T):
        r0, q0 = runs[0][t]; r1, q1 = runs[NO_QUERY_RECORDS][t]
        assert q1 is None
        assert np.array_equal(r0["match_of_feature"], r1["match_of_feature"]) and r0["n_temporal"] == r1["n_temporal"]
        if t == 0:
            assert r0["n_queries"] == 0
            continue
        prev = runs[0][t - 1][0]
        cam_of = np.repeat(np.arange(2, dtype=np.int32), prev["counts"])
        expq = pipeline.make_queries((prev["kps"], prev["desc"], prev["depth"], cam_of, prev["un_x"], prev["un_y"]), scale)
        assert q0 is not None and len(q0) == len(expq) == len(prev["kps"])
        assert q0.tobytes() == expq.tobytes()
    assert runs[0][T - 1][0]["n_temporal"] > 300

def test_extraction_is_stable_on_a_rig_of_four_threads_and_twelve_streams():
    """Forty runs of configs[3]'s loopback rig (four ranks = four host threads, three extractor handles each, one late rank, the exchange
    probe on), every rank's every extraction held against the oracle's keypoints and descriptors.  This is the load that showed wrong
    descriptor bits about once in 10^5 keypoints while k_describe multiplied freshly loaded table registers with packed f32
    instructions (round 4; csrc/Makefile: -fno-slp-vectorize) -- one in fifty runs then, so forty runs see such a fault every other time."""
    import threading
    import time
    import oracle
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline
    from multi_orb_slam_amd.dist import shard_cameras
    world, n_cams, w, h, nf, ahead, T, runs = 4, 4, 640, 480, 1000, 2, 10, 40
    frames = [{g: synth.image(g, t, w, h) for g in range(n_cams)} for t in range(T)]
    expect = [[oracle.extract(frames[t][r], nfeatures=nf) for t in range(T)] for r in range(world)]
    for run in range(runs):
        results = [[None] * T for _ in range(world)]
        errors = []

        def rank_main(r):
            try:
                mine = shard_cameras(n_cams, world, r)
                fe = pipeline.FrontEnd([m.ExtractorParams(nfeatures=nf)], w, h, rank=r, world_size=world, global_cams=mine)
                fe.fe.exchange_init_loopback(5000 + run, world, r)
                fe.native_exchange = True
                fe.fe.debug_exchange_timing(True)
                announced = 0
                for t in range(T):
                    if r == 1:
                        time.sleep(1.0e-3)
                    while announced < min(t + ahead, T - 1):
                        announced += 1
                        fe.announce([frames[announced][g] for g in mine])
                    announced = max(announced, t)
                    res = fe.step([frames[t][g] for g in mine])
                    results[r][t] = (res["kps"], res["desc"])
                    fe.fe.debug_exchange_us()
                fe.fe.exchange_shutdown()
                fe.close()
            except Exception as e:      # noqa: BLE001 -- reported by the main thread
                errors.append((r, repr(e)))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(120)
        assert not errors and not any(th.is_alive() for th in threads), errors
        for r in range(world):
            for t in range(T):
                kp, d = results[r][t]
                ek, ed = expect[r][t]
                assert kp.tobytes() == ek.tobytes(), ("keypoints", run, r, t)
                assert np.array_equal(d, ed), ("descriptors", run, r, t, np.flatnonzero((d != ed).any(axis=1))[:8].tolist())


@pytest.mark.timeout(600, method="thread")
def test_reserved_compute_units_change_nothing_but_time():
    """MORB_RESERVE_CUS=2: every extractor's queue is created with a CU mask (two units per XCD stay free for the matcher's stream:
    csrc/extractor.hip, orbx_create).  Where a workgroup runs cannot change a result: the overlapped-step and six-camera tests of this
    file once more in a child with the masks on, against the oracle."""
    import os, subprocess, sys
    env = dict(os.environ, MORB_RESERVE_CUS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "test_overlapped_steps_equal_oracle_pipeline or test_six_camera_rig or test_native_stream_loop"],
                       env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
