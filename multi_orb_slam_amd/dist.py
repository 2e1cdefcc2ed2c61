"""Multi-GPU layer: one process per GPU, cameras sharded across ranks, ONE collective per timestep.

The reference has no distributed path (single process, two cameras extracted back to back, src/Frame.cc:182-185).
Camera streams are independent until cross-camera matching, so the path shards by camera with no data-path
collective except one all-gather of the fixed-capacity descriptor blocks (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for the world_size-2 tests).  The extractor writes its descriptors straight into the
all-gather send buffer (orbx_bind_output), so there is no staging copy.
"""
import numpy as np


class DescriptorExchange:
    """All-gather of [n_cams, cap, 32] descriptor blocks + counts over torch.distributed."""

    def __init__(self, n_cams, cap, device, dist):
        import torch
        self.torch, self.dist = torch, dist
        self.n_cams, self.cap = n_cams, cap
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.send_desc = torch.zeros((n_cams, cap, 32), dtype=torch.uint8, device=device)
        self.send_kps = torch.zeros((n_cams, cap, 28), dtype=torch.uint8, device=device)
        self.recv_desc = torch.zeros((self.world * n_cams, cap, 32), dtype=torch.uint8, device=device)
        self.send_cnt = torch.zeros(n_cams, dtype=torch.int32, device=device)
        self.recv_cnt = torch.zeros(self.world * n_cams, dtype=torch.int32, device=device)

    def bind(self, extractor):
        for c in range(self.n_cams):
            extractor.bind_output(c, self.send_kps[c].data_ptr(), self.send_desc[c].data_ptr(), self.cap)

    def __call__(self, frontend, counts):
        """-> (device pointers, counts) of every camera's descriptor block of the whole rig, global camera order."""
        torch, dist = self.torch, self.dist
        self.send_cnt.copy_(torch.tensor(counts, dtype=torch.int32))
        if self.send_desc.is_cuda:
            # descriptors were produced on the extractor's stream: make sure they have landed before RCCL reads them
            from . import rt
            rt.stream_sync(frontend.stream)
        dist.all_gather_into_tensor(self.recv_desc, self.send_desc)
        dist.all_gather_into_tensor(self.recv_cnt, self.send_cnt)
        if self.send_desc.is_cuda:
            torch.cuda.current_stream().synchronize()
        cnt = self.recv_cnt.cpu().numpy().tolist()
        ptrs = [self.recv_desc[g].data_ptr() for g in range(self.world * self.n_cams)]
        return ptrs, cnt


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
