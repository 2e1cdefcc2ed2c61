#!/usr/bin/env python3
"""Probe: can two processes share ONE GPU in an RCCL communicator on this image?  (RCCL normally refuses with
"Duplicate GPU detected"; if some setting allows it, the world_size-2 path of the native exchange can be exercised on the
1-GPU test box.)"""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        t = torch.full((4,), float(rank + 1), device="cuda")
        out = torch.zeros(8, device="cuda")
        dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize()
        print("rank", rank, "OK", out.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank", rank, "FAILED", type(e).__name__, str(e)[:300], flush=True)

if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29777), nprocs=2, join=True)
