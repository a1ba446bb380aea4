"""Multi-GPU plan for the denoise hot path: one process per GPU, images are the independent
units (each image's uncond+cond CFG pair stays on one rank), so the batch is partitioned
contiguously over ranks and there is NO collective inside a step.  The only exchanges are
outside the loop: one RCCL broadcast (backend "nccl" == RCCL on ROCm, over xGMI) of the flat
fp16 weight arena at start-up, and an optional final all-gather of the latents.

The reference has no multi-GPU inference at all; this is the design of
SURVEY.md section 8(e).  Everything here also runs on CPU tensors with the gloo backend,
which is how tests/test_distributed.py covers world_size = 2.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str = None):
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # "nccl" is RCCL on ROCm; CID_DIST_BACKEND lets a single-GPU test rig run the N > 1 path over gloo
        backend = backend or os.environ.get("CID_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the `total` images owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flatten(tensors: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, List[Tuple[Tuple[int, ...], int]]]:
    """Pack fp16 tensors into one contiguous arena (16-byte aligned slots) + (shape, offset) meta."""
    meta, off = [], 0
    for t in tensors:
        assert t.dtype == torch.float16
        meta.append((tuple(t.shape), off))
        off += (t.numel() + 7) // 8 * 8
    flat = torch.zeros(off, dtype=torch.float16, device=tensors[0].device if tensors else "cpu")
    for t, (_, o) in zip(tensors, meta):
        flat[o:o + t.numel()].copy_(t.reshape(-1))
    return flat, meta


def unflatten(flat: torch.Tensor, meta) -> List[torch.Tensor]:
    out = []
    for shape, off in meta:
        n = 1
        for d in shape:
            n *= d
        out.append(flat[off:off + n].view(shape))
    return out


def layout(tensors: Sequence[torch.Tensor]) -> Tuple[List[Tuple[Tuple[int, ...], int]], int]:
    """(shape, offset) of every tensor in the flat arena (16-byte aligned slots) and the arena's size in halfs"""
    meta, off = [], 0
    for t in tensors:
        assert t.dtype == torch.float16
        meta.append((tuple(t.shape), off))
        off += (t.numel() + 7) // 8 * 8
    return meta, off


def broadcast_weights(named: Dict[str, torch.Tensor], device, src: int = 0, bucket_bytes: int = 256 << 20,
                      single_rank_too: bool = False) -> Dict[str, torch.Tensor]:
    """Rank `src` owns `named` (others may pass {}); afterwards every other rank holds views into one flat arena with
    identical contents, rank `src` keeps its own tensors.  The arena travels in buckets of whole tensors of about
    `bucket_bytes` (large transfers: an xGMI ring is per-link bound, so few big collectives beat one per tensor): the
    source stages ONE bucket at a time instead of a second copy of the whole arena, receivers write straight into
    their slice of the arena.  ``single_rank_too``: run the staging + collective calls even in a one-rank group (a rig
    with one GPU cannot host two RCCL ranks; this is how the RCCL code path itself gets executed there)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (single_rank_too and dist.is_initialized()):
        return named
    rank = dist.get_rank()
    if rank == src:
        keys = sorted(named)
        meta, numel = layout([named[k] for k in keys])
        header = [keys, meta, numel]
    else:
        header = [None, None, None]
    dist.broadcast_object_list(header, src=src)
    keys, meta, numel = header
    flat = torch.zeros(numel, dtype=torch.float16, device=device) if rank != src else None
    ends = [off for _, off in meta[1:]] + [numel]          # end of tensor i = start of tensor i + 1 (padding included)
    i, n = 0, len(meta)
    while i < n:
        lo = meta[i][1]
        j = i + 1
        while j < n and (ends[j] - lo) * 2 <= bucket_bytes:
            j += 1
        hi = ends[j - 1]
        if rank == src:
            t0 = named[keys[i]]
            buf = torch.zeros(hi - lo, dtype=torch.float16, device=t0.device)
            for k in range(i, j):
                t = named[keys[k]]
                o = meta[k][1] - lo
                buf[o:o + t.numel()].copy_(t.reshape(-1))
            dist.broadcast(buf, src=src)
            del buf
        else:
            dist.broadcast(flat[lo:hi], src=src)
        i = j
    if rank == src:
        return named
    return dict(zip(keys, unflatten(flat, meta)))


def all_gather_latents(local: torch.Tensor, total: int, single_rank_too: bool = False) -> torch.Tensor:
    """Optional: collect every rank's [b_r, 4, h, w] latents (32 KB / image) on all ranks."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (single_rank_too and dist.is_initialized()):
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]].copy_(local)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)
