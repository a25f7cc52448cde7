"""Utterance sharding across the GPUs of one node + the single waveform gather.

Utterances are independent (batch-1 semantics), so ranks never exchange data on the synthesis path; the
only collective is the final gather of finished waveforms to rank 0.  On GPUs that gather runs INSIDE libzvx
(`zvx_comm_gather`: grouped ncclSend / ncclRecv on RCCL over xGMI, every peer->root transfer on its own link; see
`Context.comm_init` / `Context.comm_gather` in _lib.py) -- no torch in the data path.  This module holds the
transport-independent pieces: the shard arithmetic, and a host-memory gather over a `torch.distributed` group
(gloo) that the CPU tests and CPU-side tooling use.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition: rank r owns [start, stop); blocks differ by at most one item."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_waveforms(local_wav, local_len, dst=0, group=None):
    """Host/gloo transport.  local_wav [B_local, N] (same B_local, N on every rank), local_len [B_local] tensors.
    Returns freshly allocated (wav [world*B_local, N], lens [world*B_local]) on group rank ``dst``, (None, None) elsewhere.
    The collective sequence is fixed (two gathers on every rank): nothing is caught or retried, an error on any rank
    surfaces as that rank's exception."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local_wav, local_len
    gdst = dist.get_global_rank(group, dst) if group is not None else dst          # dist.gather takes a GLOBAL rank
    B = local_wav.shape[0]
    full_w = full_l = wavs = lens = None
    if rank == dst:
        full_w = torch.empty((world * B,) + tuple(local_wav.shape[1:]), dtype=local_wav.dtype, device=local_wav.device)
        full_l = torch.empty((world * B,), dtype=local_len.dtype, device=local_len.device)
        wavs, lens = list(full_w.split(B)), list(full_l.split(B))       # views: the peers' rows land in place
    dist.gather(local_wav, wavs, dst=gdst, group=group)
    dist.gather(local_len, lens, dst=gdst, group=group)
    return full_w, full_l
