"""world_size-2 gloo run of the multi-GPU plan on CPU: weight-arena broadcast, image sharding,
final latent all-gather.  No collective exists inside a denoise step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from consistentid_amd import distributed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        named = {}
        if rank == 0:
            g = torch.Generator().manual_seed(0)
            named = {"b.w": torch.randn(5, 7, generator=g).half(), "a.w": torch.randn(33, generator=g).half()}
        got = distributed.broadcast_weights(named, "cpu", bucket_bytes=64)     # 33 + 35 halfs: two buckets
        g = torch.Generator().manual_seed(0)
        want = {"b.w": torch.randn(5, 7, generator=g).half(), "a.w": torch.randn(33, generator=g).half()}
        assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want)
        lo, hi = distributed.shard_range(total, rank, world)
        # every rank derives its images from (seed + global image index): no scatter needed
        local = torch.stack([torch.full((4, 2, 2), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 4, 2, 2)
        allv = distributed.all_gather_latents(local, total)
        assert allv.shape[0] == total and [float(allv[i, 0, 0, 0]) for i in range(total)] == [float(i) for i in range(total)]
    finally:
        dist.destroy_process_group()


def test_gloo_world_size_2():
    mp.spawn(_worker, args=(2, _free_port(), 5), nprocs=2, join=True)
