"""world_size-2 gloo test of the multi-GPU path: utterance sharding + the single waveform gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_synth(first, n, N):
    # stands in for the GPU synthesis: row content is a pure function of the global utterance index
    rows = torch.arange(first, first + n, dtype=torch.float32)[:, None] * 1000.0 + torch.arange(N, dtype=torch.float32)[None, :]
    lens = torch.arange(first, first + n, dtype=torch.int32) + 1
    return rows, lens


def _worker(rank, world, port, B_total, N, q):
    sys.path.insert(0, ROOT)
    from zerovox_amd.dist import gather_waveforms, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(B_total, rank, world)
    wav, lens = _fake_synth(a, b - a, N)
    gw, gl = gather_waveforms(wav, lens, dst=0)
    if rank == 0:
        q.put((gw.numpy(), gl.numpy()))
    else:
        assert gw is None and gl is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2_equals_single_rank():
    B_total, N, world = 8, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, B_total, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    gw, gl = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref_w, ref_l = _fake_synth(0, B_total, N)
    assert np.array_equal(gw, ref_w.numpy()) and np.array_equal(gl, ref_l.numpy())
