#!/usr/bin/env python3
"""One RANK PROCESS of the multi-process rehearsal of the N > 1 job.  TEST INFRASTRUCTURE.

    python tests/mp_rank.py <world> <rank> <port> <rig_cams> <W> <H> <nf> <steps> <ahead> <device> [die_at=<t>:<rank>] [photo=<name>]
                            [aheads=a0,a1,..] [noise_at=<t>:<cam>] [slow=<rank>:<ms>] [verbose=1]

Every rank is a fresh process (started by tests/test_gpu_multiprocess.py or by hand) that owns rig_cams / world cameras of ONE rig of
overlapping cameras sliding over a photograph (tests/natural.py), joins a gloo group on 127.0.0.1 -- the control plane: it carries the
64-byte IPC handles once and the final verdicts --, sets up the peer transport (orbf_exchange_peer_*: direct writes into the other
ranks' arenas) and runs `steps` native steps with `ahead` timesteps announced.  Every step is held against the oracle: this rank's
keypoints, descriptors, stereo and temporal matches, and the rig-wide top-2 of its features against the cameras of ALL OTHER ranks (whose
descriptors the checker computes itself).  All ranks may share one GPU (device 0 for everybody): that is the rehearsal a 1-GPU box allows.
die_at=<t>:<rank>: that rank exits without a word before step t -- the others must come back from their step with ORB_E_TIMEOUT.
Exit status: 0 ok, 3 = the expected timeout was reported (die_at runs), anything else = failure."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world, rank, port, rig_cams, W, H, nf, steps, ahead, device = (int(x) for x in sys.argv[1:11])
    opts = dict(a.split("=", 1) for a in sys.argv[11:])
    die_t, die_rank = (int(x) for x in opts["die_at"].split(":")) if "die_at" in opts else (-1, -1)
    photo = opts.get("photo", "china")
    if "aheads" in opts:            # look-ahead per rank (ranks that run ahead by different amounts ship their blocks at different moments)
        ahead = [int(x) for x in opts["aheads"].split(",")][rank]
    noise_t, noise_cam = (int(x) for x in opts["noise_at"].split(":")) if "noise_at" in opts else (-1, -1)
    slow_rank, slow_ms = (int(x) for x in opts["slow"].split(":")) if "slow" in opts else (-1, 0)   # that rank sleeps before every step
    import datetime
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    import multi_orb_slam_amd as m
    from multi_orb_slam_amd import pipeline, _lib
    from multi_orb_slam_amd.dist import shard_cameras
    from oracle_pipeline import OracleFrontEnd, assert_same_step
    import natural, oracle
    per = rig_cams // world
    mine = shard_cameras(rig_cams, world, rank)
    params = [m.ExtractorParams(nfeatures=nf)] * per
    fe = pipeline.FrontEnd(params, W, H, device=device, rank=rank, world_size=world, global_cams=mine)
    ok = fe.enable_peer_exchange(dist)
    assert ok, "peer exchange could not be set up"
    assert fe.fe.exchange_world == world and fe.fe.exchange_placement == 3
    ofe = OracleFrontEnd(params, W, H, mine)
    frames = [natural.rig(photo, t, W, H, n_cams=rig_cams) for t in range(steps)]
    if noise_t >= 0:                # one camera sees noise in one step: more candidates than the device quadtree takes (MORB_OCT_MAX_KEYS lowered by
        from multi_orb_slam_amd import synth   # the test) -> that rank's step is redone on the host path and EVERY rank ships its block a second time
        r_ = synth.hash32(np.arange(W * H, dtype=np.uint64) + np.uint64(1000 * noise_t + 17 * noise_cam))
        frames[noise_t][noise_cam] = (r_ % 256).astype(np.uint8).reshape(H, W)
    announced = 0
    t_step = []
    for t in range(steps):
        if rank == die_rank and t == die_t:
            os._exit(0)                      # no shutdown, no goodbye: a dead rank
        while announced < min(t + ahead, steps - 1):
            announced += 1
            fe.announce([frames[announced][g] for g in mine])
        announced = max(announced, t)
        if rank == slow_rank:
            time.sleep(slow_ms / 1000.0)
        t0 = time.time()
        try:
            got = fe.step([frames[t][g] for g in mine])
        except _lib.OrbError as e:
            dt = time.time() - t0
            print("rank %d step %d: %s (after %.1f s)" % (rank, t, e, dt), flush=True)
            if die_t >= 0 and e.code == _lib.ORB_E_TIMEOUT and dt < 30.0:
                os._exit(3)                  # the expected outcome of a die_at run (no collective shutdown with a dead peer)
            os._exit(4)
        t_step.append(time.time() - t0)
        if "verbose" in opts:
            print("rank %d step %d done in %.3f s, redos so far %d" % (rank, t, t_step[-1], fe.fe.debug_exchange_redos()), flush=True)
        # the checker: the other cameras' descriptors from the oracle's own extraction
        others = {g: oracle.extract(frames[t][g], nfeatures=nf)[1] for g in range(rig_cams) if g not in mine}
        exp = ofe.step([frames[t][g] for g in mine],
                       other_descs=lambda c: [others[g] if g in others else exp_own[g] for g in range(rig_cams) if g != mine[c]],
                       on_extracted=lambda per_cam: exp_own.update({g: per_cam[i][1] for i, g in enumerate(mine)}))
        assert_same_step(got, exp)
        assert got["rig_counts"] == [len(others[g]) if g in others else len(exp_own[g]) for g in range(rig_cams)]
    verdicts = [None] * world
    dist.all_gather_object(verdicts, (rank, got["counts"], got["n_temporal"], got["n_cross"], fe.fe.debug_exchange_redos()))
    fe.fe.exchange_shutdown()
    fe.close()
    if rank == 0:
        print("mp_rank: %d ranks x %d camera(s) %dx%d @%d, %d steps, %d ahead: every step of every rank bit-exact vs the oracle; last step %s; "
              "median step %.2f ms" % (world, per, W, H, nf, steps, ahead, verdicts, 1e3 * float(np.median(t_step))), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


exp_own = {}
if __name__ == "__main__":
    sys.exit(main())
