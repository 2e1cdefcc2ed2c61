"""Multi-GPU layer: one process per GPU, cameras sharded across ranks, ONE collective per timestep.

The reference has no distributed path (single process, two cameras extracted back to back, src/Frame.cc:182-185).
Camera streams are independent until cross-camera matching, so the path shards by camera with no data-path
collective except one all-gather of the fixed-capacity descriptor blocks (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for the world_size-2 tests).  The send buffer IS the front end's frame (its describe kernel wrote
the descriptors there), so there is no staging copy and the next timestep's extraction may already be running.
"""
import numpy as np


class _DeviceBlock:
    """Zero-copy view of a native HBM block for torch (the __cuda_array_interface__ protocol)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class DescriptorExchange:
    """ONE all-gather per timestep over torch.distributed: every rank ships its front end's export block (the merged
    descriptors of its cameras + the per-camera counts in a trailer, orbf_export_block) and matches its own features against
    the gathered blocks of the whole rig (orbm_cross_top2_gathered).  Nothing is staged and no count is exchanged separately."""

    def __init__(self, device, dist):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.recv = None
        self._send = {}

    def __call__(self, frontend):
        """frontend: pipeline.FrontEnd after a step -> (best_idx, best_dist, second_dist, counts of all cameras)."""
        torch, dist = self.torch, self.dist
        ptr, nbytes, rows = frontend.fe.export_block()
        send = self._send.get(ptr)
        if send is None:      # two blocks alternate (the front end's double-buffered frames): wrapped once each
            send = self._send[ptr] = torch.as_tensor(_DeviceBlock(ptr, nbytes), device=self.device)
        if self.recv is None or self.recv.numel() != self.world * nbytes:
            self.recv = torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.device)
            self.recv_ptr = self.recv.data_ptr()
        # the block was complete before the step returned (the matcher waited on the extractor's event), so RCCL may read
        # it on torch's stream right away; the matcher's kernels then wait for the collective on the host
        dist.all_gather_into_tensor(self.recv, send)
        torch.cuda.current_stream().synchronize()
        return frontend.mt.cross_top2_gathered(self.recv_ptr, self.world, nbytes, rows, frontend.n_cams, self.rank)


def shard_cameras(n_cameras, world_size, rank):
    """Camera c is owned by rank c // ceil(n/world): contiguous blocks keep the global camera order == rank order."""
    per = (n_cameras + world_size - 1) // world_size
    return list(range(rank * per, min(n_cameras, (rank + 1) * per)))


def gather_numpy(dist, per_cam, cap):
    """CPU/gloo mirror of DescriptorExchange used by the world_size-2 tests: returns {global_cam: descriptors}."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n_cams = len(per_cam)
    send = torch.zeros((n_cams, cap, 32), dtype=torch.uint8)
    cnt = torch.zeros(n_cams, dtype=torch.int32)
    for c, (k, d) in enumerate(per_cam):
        send[c, :len(d)] = torch.from_numpy(np.ascontiguousarray(d))
        cnt[c] = len(d)
    recv = torch.zeros((world * n_cams, cap, 32), dtype=torch.uint8)
    rcnt = torch.zeros(world * n_cams, dtype=torch.int32)
    dist.all_gather_into_tensor(recv, send)
    dist.all_gather_into_tensor(rcnt, cnt)
    return {g: recv[g, :int(rcnt[g])].numpy() for g in range(world * n_cams)}
