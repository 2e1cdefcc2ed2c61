"""N > 1 path on CPU: world_size-2 gloo processes run the camera sharding + descriptor all-gather logic of
multi_orb_slam_amd.dist and check the cross-camera top-2 of every rank against a single-process computation.
(The HIP kernels need a GPU; here the per-rank compute is the oracle, the thing under test is the exchange.)"""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, ret):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from multi_orb_slam_amd import synth
    from multi_orb_slam_amd.dist import shard_cameras, gather_numpy
    n_cameras, cap = 4, 1500                 # rows per rank block (both cameras of a rank share it)
    mine = shard_cameras(n_cameras, world, rank)
    assert mine == [2 * rank, 2 * rank + 1]
    # ragged per-camera descriptor sets (camera g has 300 + 100*g descriptors; the ranks' blocks are unevenly filled)
    per_cam = [(np.zeros(300 + 100 * g, oracle.KP_DTYPE), synth.descriptors(300 + 100 * g, 1000 + g)) for g in mine]
    everyone = gather_numpy(dist, per_cam, cap)
    assert sorted(everyone) == [0, 1, 2, 3]
    out = {}
    for c, g in enumerate(mine):
        refs = np.concatenate([everyone[o] for o in range(n_cameras) if o != g])
        out[g] = oracle.bf_top2(per_cam[c][1], refs)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_descriptor_exchange_matches_single_process():
    import oracle
    from multi_orb_slam_amd import synth
    world, port = 2, 29000 + (os.getpid() % 2000)
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    descs = {g: synth.descriptors(300 + 100 * g, 1000 + g) for g in range(4)}
    for rank in range(world):
        for g, (bi, bd, sd) in ret[rank].items():
            refs = np.concatenate([descs[o] for o in range(4) if o != g])
            ebi, ebd, esd = oracle.bf_top2(descs[g], refs)
            assert np.array_equal(bi, ebi) and np.array_equal(bd, ebd) and np.array_equal(sd, esd)


def test_shard_cameras_covers_every_camera_once():
    import pytest
    from multi_orb_slam_amd.dist import shard_cameras
    for n in (1, 2, 4, 7, 8):
        for w in (1, 2, 4, 8):
            if n % w:      # the exchange needs equal shards: a ragged rig is refused unless asked for explicitly
                with pytest.raises(ValueError):
                    shard_cameras(n, w, 0)
            owned = [c for r in range(w) for c in shard_cameras(n, w, r, allow_ragged=True)]
            assert owned == list(range(n))
            if n % w == 0:
                assert all(len(shard_cameras(n, w, r)) == n // w for r in range(w))


def test_unpack_refuses_a_corrupt_trailer():
    import pytest
    from multi_orb_slam_amd import synth
    from multi_orb_slam_amd.dist import pack_export_block, unpack_gathered
    cap = 64
    a = pack_export_block([synth.descriptors(10, 1), synth.descriptors(20, 2)], cap)
    b = pack_export_block([synth.descriptors(5, 3), synth.descriptors(7, 4)], cap)
    got = unpack_gathered(np.concatenate([a, b]), 2, cap, 2)
    assert [len(got[g]) for g in range(4)] == [10, 20, 5, 7]
    b[cap * 32:cap * 32 + 4] = np.array([60], np.int32).view(np.uint8)     # 60 + 7 rows do not fit 64
    with pytest.raises(ValueError):
        unpack_gathered(np.concatenate([a, b]), 2, cap, 2)
    b[cap * 32:cap * 32 + 4] = np.array([-3], np.int32).view(np.uint8)
    with pytest.raises(ValueError):
        unpack_gathered(np.concatenate([a, b]), 2, cap, 2)
