"""Utterance sharding across the GPUs of one node + the single waveform gather.

Utterances are independent (batch-1 semantics), so ranks never exchange data on the synthesis path; the
only collective is the final gather of finished waveforms to rank 0 (RCCL over xGMI: every peer->root
transfer rides its own link).  One process per GPU, torch.distributed as plumbing only.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition: rank r owns [start, stop); blocks differ by at most one item."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_waveforms(local_wav, local_len, dst=0, group=None):
    """local_wav [B_local, N] (same B_local, N on every rank), local_len [B_local] int32 tensors on the
    rank's device.  Returns (wav [B, N], lens [B]) on rank ``dst`` (None elsewhere)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local_wav, local_len
    if rank == dst:
        wavs = [torch.empty_like(local_wav) for _ in range(world)]
        lens = [torch.empty_like(local_len) for _ in range(world)]
    else:
        wavs = lens = None
    try:
        dist.gather(local_wav, wavs, dst=dst, group=group)
        dist.gather(local_len, lens, dst=dst, group=group)
    except (RuntimeError, NotImplementedError):
        # a backend without gather: fall back to all_gather (every rank receives all rows, only dst keeps them)
        wavs = [torch.empty_like(local_wav) for _ in range(world)]
        lens = [torch.empty_like(local_len) for _ in range(world)]
        dist.all_gather(wavs, local_wav, group=group)
        dist.all_gather(lens, local_len, group=group)
    if rank == dst:
        return torch.cat(wavs, 0), torch.cat(lens, 0)
    return None, None
