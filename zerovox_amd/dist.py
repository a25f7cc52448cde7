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
    rank's device.  Returns (wav [B, N], lens [B]) on rank ``dst`` (None elsewhere).  The returned tensors are a landing
    buffer that the next call with the same shapes overwrites: consume (or clone) them first."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local_wav, local_len
    B = local_wav.shape[0]

    def landing():
        # the peers' rows land directly in one [world*B, N] buffer (views as the gather list: no concatenation pass);
        # the buffer is reused across calls of the same shape
        key = (tuple(local_wav.shape), local_wav.dtype, str(local_wav.device), world)
        if _cache.get("key") != key:
            _cache["key"] = key
            _cache["wav"] = torch.empty((world * B,) + tuple(local_wav.shape[1:]), dtype=local_wav.dtype, device=local_wav.device)
            _cache["len"] = torch.empty((world * B,), dtype=local_len.dtype, device=local_len.device)
        return _cache["wav"], _cache["len"]

    full_w = full_l = None
    if rank == dst:
        full_w, full_l = landing()
        wavs, lens = list(full_w.split(B)), list(full_l.split(B))
    else:
        wavs = lens = None
    try:
        dist.gather(local_wav, wavs, dst=dst, group=group)
        dist.gather(local_len, lens, dst=dst, group=group)
    except (RuntimeError, NotImplementedError):
        # a backend without gather: fall back to all_gather (every rank receives all rows, only dst keeps them)
        full_w, full_l = landing()
        dist.all_gather(list(full_w.split(B)), local_wav, group=group)
        dist.all_gather(list(full_l.split(B)), local_len, group=group)
    if rank == dst:
        return full_w, full_l
    return None, None


_cache: dict = {}
