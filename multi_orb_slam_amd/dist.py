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

    def _gather(self, frontend, block=None):
        torch, dist = self.torch, self.dist
        ptr, nbytes, rows = block if block is not None else frontend.fe.export_block()
        send = self._send.get(ptr)
        if send is None:      # a few blocks alternate (the front end's rotating frames): wrapped once each
            send = self._send[ptr] = torch.as_tensor(_DeviceBlock(ptr, nbytes), device=self.device)
        if self.recv is None or self.recv.numel() != self.world * nbytes:
            self.recv = torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.device)
            self.recv_ptr = self.recv.data_ptr()
        dist.all_gather_into_tensor(self.recv, send)
        return nbytes, rows

    def __call__(self, frontend):
        """After a completed step -> (best_idx, best_dist, second_dist, counts of all cameras)."""
        # the block was complete before the step returned (the matcher waited on the extractor's event), so RCCL may read
        # it on torch's stream right away; the matcher's stream is ordered behind the collective on the device
        nbytes, rows = self._gather(frontend)
        if self.recv.is_cuda:
            frontend.mt.wait_for_stream(self.torch.cuda.current_stream().cuda_stream)
        return frontend.mt.cross_top2_gathered(self.recv_ptr, self.world, nbytes, rows, frontend.n_cams, self.rank)

    def gather_ahead(self, frontend, block):
        """Before the step is even begun (block = FrontEnd.peek_block(images), final already): start the collective now."""
        self._ahead = self._gather(frontend, block)

    def enqueue(self, frontend):
        """Between FrontEnd begin and end, when begin reported the export block ready: the collective (unless gather_ahead
        started it) and the gathered matching are enqueued next to the step's own matching; collect() after end() returns
        what __call__ would."""
        ahead = getattr(self, "_ahead", None)
        self._ahead = None
        nbytes, rows = ahead if ahead is not None else self._gather(frontend)
        frontend.mt.cross_top2_gathered_enqueue(self.recv_ptr, self.world, nbytes, rows, frontend.n_cams, self.rank,
                                                self.torch.cuda.current_stream().cuda_stream if self.recv.is_cuda else None)

    def collect(self, frontend, views=False):
        return frontend.mt.cross_top2_gathered_collect_views() if views else frontend.mt.cross_top2_gathered_collect()


def shard_cameras(n_cameras, world_size, rank, allow_ragged=False):
    """Camera c is owned by rank c // ceil(n/world): contiguous blocks keep the global camera order == rank order.
    The exchange's wire format (export block + count trailer, k_repack_gathered) indexes cameras as rank * cams_per_rank + c,
    i.e. it needs the SAME number of cameras on every rank: a rig that does not divide evenly is refused unless the caller
    asks for the ragged split explicitly (bookkeeping only -- such a split cannot go through the exchange)."""
    per = (n_cameras + world_size - 1) // world_size
    if n_cameras % world_size and not allow_ragged:
        raise ValueError("%d cameras do not shard evenly over %d ranks (the descriptor exchange needs equal shards)" % (n_cameras, world_size))
    return list(range(rank * per, min(n_cameras, (rank + 1) * per)))


BLOCK_TRAILER = 256  # bytes behind the descriptor rows of an export block: int32 per-camera counts (orbf_export_block)


def pack_export_block(per_cam_desc, cap_rows):
    """numpy restatement of the block orbf_export_block hands out: the rank's cameras packed back to back in cap_rows rows of
    32 bytes (unused rows left as they are), then the count trailer."""
    blk = np.full(cap_rows * 32 + BLOCK_TRAILER, 0xA5, np.uint8)  # unused rows / trailer bytes: arbitrary content
    packed = np.concatenate([np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in per_cam_desc]) if per_cam_desc else np.zeros((0, 32), np.uint8)
    assert len(packed) <= cap_rows
    blk[:packed.size] = packed.reshape(-1)
    blk[cap_rows * 32:cap_rows * 32 + 4 * len(per_cam_desc)] = np.array([len(d) for d in per_cam_desc], np.int32).view(np.uint8)
    return blk


def unpack_gathered(gathered, world, cap_rows, cams_per_rank):
    """numpy restatement of what k_repack_gathered reads out of the all-gathered blocks: {global camera: descriptors}."""
    block_bytes = cap_rows * 32 + BLOCK_TRAILER
    out = {}
    for r in range(world):
        blk = gathered[r * block_bytes:(r + 1) * block_bytes]
        counts = blk[cap_rows * 32:cap_rows * 32 + 4 * cams_per_rank].view(np.int32)
        if (counts < 0).any() or int(counts.sum()) > cap_rows:
            raise ValueError("rank %d's trailer holds counts %s that do not fit its %d rows" % (r, counts.tolist(), cap_rows))
        rows = blk[:cap_rows * 32].reshape(cap_rows, 32)
        off = 0
        for c in range(cams_per_rank):
            out[r * cams_per_rank + c] = rows[off:off + int(counts[c])].copy()
            off += int(counts[c])
    return out


def gather_numpy(dist, per_cam, cap_rows):
    """CPU/gloo mirror of DescriptorExchange used by the world_size-2 tests: the same wire format (one export block per
    rank, one all-gather), packed and unpacked by the numpy restatements above.  Returns {global_cam: descriptors}."""
    import torch
    world = dist.get_world_size()
    send = torch.from_numpy(pack_export_block([d for (_k, d) in per_cam], cap_rows))
    recv = torch.zeros(world * send.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)
    return unpack_gathered(recv.numpy(), world, cap_rows, len(per_cam))
