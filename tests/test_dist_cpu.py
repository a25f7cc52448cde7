"""world_size-2 tests of the multi-GPU path on CPU.

1. `shard_range` + the host/gloo `gather_waveforms` transport (zerovox_amd/dist.py).
2. bench.py's OWN control flow for N > 1 -- env parsing, the RCCL-id rendezvous through the TCPStore, the double-buffered
   synthesize -> gather step, barrier / MAX-over-ranks timing, rank 0's JSON line -- driven through `bench.main` with a stub
   context: the stub implements the libzvx calls bench.py makes (synthesize into "device" buffers, comm_init / comm_gather /
   comm_barrier / comm_max, dev_alloc / dev_to_host) on host memory with gloo as the transport, so everything except the
   kernels and RCCL itself is the code the driver runs on 8 GPUs.
"""
import json
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


def _spawn(target, world, extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=target, args=(r, world, port) + extra + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return out


def test_shard_and_gather_world2_equals_single_rank():
    B_total, N, world = 8, 16, 2
    gw, gl = _spawn(_worker, world, (B_total, N))
    ref_w, ref_l = _fake_synth(0, B_total, N)
    assert np.array_equal(gw, ref_w.numpy()) and np.array_equal(gl, ref_l.numpy())


# ------------------------------------------------------------------------------------------------
# bench.py control flow with a stub context
# ------------------------------------------------------------------------------------------------
class StubContext:
    """Host-memory stand-in for zerovox_amd._lib.Context with gloo as the gather transport."""
    hidden, n_mels, hop = 528, 80, 256

    def __init__(self):
        self._bufs, self._next, self.calls = {}, 1, []
        self.rank, self.world = 0, 1

    # device memory
    def dev_alloc(self, nbytes):
        p = self._next; self._next += 1
        self._bufs[p] = np.zeros(nbytes, np.uint8)
        return p

    def dev_to_host(self, ptr, shape, dtype):
        return self._bufs[ptr][: int(np.prod(shape)) * np.dtype(dtype).itemsize].view(dtype).reshape(shape).copy()

    # synthesis: row b of the shard = (first utterance id + b + 1) / 1000 everywhere (a pure function of the global index)
    def synthesize(self, ph, pu, T, spk, dur, pad_to, want_mel=True, wav_device_ptr=None, wav_stride=None, no_sync=False, pcm16=False):
        B = ph.shape[0]
        ids = self._first + np.arange(B, dtype=np.float32)
        self._bufs[wav_device_ptr].view(np.float32).reshape(B, wav_stride)[:] = ((ids + 1) / 1000.0)[:, None]
        self.calls.append(("synthesize", wav_device_ptr))

    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    def comm_init(self, cid, rank, world):
        assert world == 1 or cid == bytes(range(128)), "every rank must receive rank 0's id"
        self.rank, self.world = rank, world
        if world > 1:
            # under torchrun the elastic agent hosts the store (env://); spawned by hand, rank 0 hosts one next to bench.py's
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
            dist.init_process_group("gloo", rank=rank, world_size=world,
                                    init_method="env://" if agent else f"tcp://127.0.0.1:{int(os.environ['MASTER_PORT']) + 1}")

    def comm_gather(self, local_ptr, nbytes, recv_ptr, root=0, no_sync=False):
        local = torch.from_numpy(self._bufs[local_ptr][:nbytes])
        if self.rank == root:
            parts = list(torch.from_numpy(self._bufs[recv_ptr])[: self.world * nbytes].split(nbytes))
            dist.gather(local, parts, dst=root)
        else:
            dist.gather(local, None, dst=root)
        self.calls.append(("gather", local_ptr))

    def comm_barrier(self):
        if self.world > 1:
            dist.barrier()

    def comm_info(self):
        # what zvx_comm_info reports, with gloo as the transport: a SUM of ones and a MAX over per-rank device slots
        ones = torch.tensor([1.0], dtype=torch.float64); slots = torch.zeros(self.world, dtype=torch.float64)
        slots[self.rank] = 0x0500 + self.rank + 1
        if self.world > 1:
            dist.all_reduce(ones, op=dist.ReduceOp.SUM); dist.all_reduce(slots, op=dist.ReduceOp.MAX)
        return {"world": self.world, "comm_count": self.world, "version_code": 0, "version": "stub", "ranks_seen": int(ones.item()),
                "device_pci": [f"stub:{int(v) - 1:04x}" for v in slots.tolist()]}

    def comm_max(self, v):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # profiling surface (nothing to report)
    def set_int(self, *a): pass
    def get_int(self, *a): return -1
    def reset_stats(self): pass
    def kernel_stats(self): return []
    def tag_stats(self): return []
    def stage_times(self): return {}
    def sync(self): pass

    def close(self):
        if self.world > 1:
            dist.destroy_process_group()


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import contextlib, io
    import bench
    from zerovox_amd import config as zcfg
    stub = StubContext()
    stub._first = rank * 4

    def factory(args, local_rank):
        assert local_rank == rank
        return stub, (zcfg.medium_modelcfg("styletts"), None, None, None)

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "4", "--phonemes", "8", "--profile", "0",
                         "--no-cpu-baseline"], ctx_factory=factory)
    assert rc == 0
    # double buffering: consecutive steps alternate between the two waveform buffers, every step is followed by its gather
    synth = [c for c in stub.calls if c[0] == "synthesize"]
    gath = [c for c in stub.calls if c[0] == "gather"]
    assert len(synth) == 4 and len(gath) == 4 and [s[1] for s in synth] == [g[1] for g in gath]
    assert synth[0][1] != synth[1][1] and synth[0][1] == synth[2][1]
    if rank == 0:
        q.put(buf.getvalue())
    else:
        assert buf.getvalue() == ""                     # only rank 0 prints


def test_bench_control_flow_world2_with_stub_context():
    out = _spawn(_bench_worker, 2, ())
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1                              # ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["unit"] == "samples/s"
    assert r["config"]["global_batch"] == 8 and r["config"]["samples_per_utt"] == 8 * 7 * 256
    assert r["output_ok"] is True                       # rank 0 checked: its own rows landed first, rank 1's rows are present
    total = 2 * 4 * r["config"]["samples_per_utt"] * 3
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * 3 - total) < 1e-6 * total
    assert r["higher_is_better"] is True and r["vs_baseline"] is None
    # the N > 1 line says who the communicator saw (zvx_comm_info): both ranks, each with its own device
    assert r["rccl"]["ranks_seen"] == 2 and r["rccl"]["comm_count"] == 2 and len(set(r["rccl"]["device_pci"])) == 2


def test_bench_in_flight_two_contexts_round_robin(capsys):
    """bench.py --in-flight 2 (one GPU): steps alternate over two contexts, each double-buffers its own waveform rows, every
    context is fenced, and the line counts every step once; with more than one rank the option is refused."""
    import bench
    from zerovox_amd import config as zcfg
    made = []

    def factory(args, local_rank):
        c = StubContext(); c._first = 0
        made.append(c)
        return c, (zcfg.medium_modelcfg("styletts"), None, None, None)

    old = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        rc = bench.main(["--in-flight", "2", "--steps", "4", "--warmup", "2", "--batch", "4", "--phonemes", "8", "--profile", "0", "--no-cpu-baseline"],
                        ctx_factory=factory)
    finally:
        os.environ.update({k: v for k, v in old.items() if v is not None})
    assert rc == 0 and len(made) == 2
    a = [c[1] for c in made[0].calls if c[0] == "synthesize"]
    b = [c[1] for c in made[1].calls if c[0] == "synthesize"]
    assert len(a) == 3 and len(b) == 3                      # 6 steps, round robin
    assert a[0] != a[1] and a[0] == a[2] and b[0] != b[1]   # each context alternates between ITS two buffers
    r = json.loads([l for l in capsys.readouterr().out.splitlines() if l.strip()][-1])
    assert r["steps"] == 4 and r["config"]["in_flight"] == 2 and r["output_ok"] is True
    total = 4 * r["config"]["samples_per_utt"] * 4
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * 4 - total) < 1e-6 * total


def test_bench_set_overrides_reach_every_context_and_the_line(capsys):
    """bench.py --set KEY=INT (repeatable): zvx_set_int on every context before the first step, echoed in config.overrides."""
    import bench
    from zerovox_amd import config as zcfg
    made = []

    class Rec(StubContext):
        def __init__(self):
            super().__init__(); self.sets = []
        def set_int(self, k, v): self.sets.append((k, v))

    def factory(args, local_rank):
        c = Rec(); c._first = 0
        made.append(c)
        return c, (zcfg.medium_modelcfg("styletts"), None, None, None)

    old = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        rc = bench.main(["--in-flight", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--phonemes", "8", "--profile", "0", "--no-cpu-baseline",
                         "--set", "voc_overlap_maxb=0", "--set", "norm_fuse_maxb=7"], ctx_factory=factory)
    finally:
        os.environ.update({k: v for k, v in old.items() if v is not None})
    assert rc == 0 and len(made) == 2
    for c in made:
        assert ("voc_overlap_maxb", 0) in c.sets and ("norm_fuse_maxb", 7) in c.sets
    r = json.loads([l for l in capsys.readouterr().out.splitlines() if l.strip()][-1])
    assert r["config"]["overrides"] == {"voc_overlap_maxb": 0, "norm_fuse_maxb": 7}


def test_bench_under_torchrun_with_stub_context():
    """The driver's own launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...): env parsing, the TCPStore rendezvous against the elastic agent's store, one JSON line."""
    import subprocess
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_stub_main.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "3", "--phonemes", "8", "--profile", "0", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["config"]["global_batch"] == 6 and r["output_ok"] is True


def test_comm_id_rendezvous_world8_over_plain_sockets(monkeypatch):
    """bench.py's RCCL-id exchange (no torch, no store): rank 0 serves the 128-byte id to 7 late / early clients, also when
    the first rendezvous port is already taken by something else."""
    import socket, threading, time
    import bench
    blocker = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    blocker.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    blocker.bind(("127.0.0.1", 0))
    base = blocker.getsockname()[1]                       # the first candidate port is occupied (and never answers with an id)
    blocker.listen(16)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("ZVX_RDZV_PORT", str(base))
    cid = bytes(range(128))
    got = {}

    def client(r, delay):
        time.sleep(delay)
        got[r] = bench.exchange_comm_id(r, 8, lambda: b"")
    ths = [threading.Thread(target=client, args=(r, 0.05 * (r % 3))) for r in range(1, 8)]
    for t in ths[:3]:
        t.start()                                          # three clients are up before rank 0 listens
    time.sleep(0.3)
    srv = threading.Thread(target=lambda: got.__setitem__(0, bench.exchange_comm_id(0, 8, lambda: cid)))
    srv.start()
    for t in ths[3:]:
        t.start()
    for t in ths + [srv]:
        t.join(60)
    blocker.close()
    assert sorted(got) == list(range(8)) and all(got[r] == cid for r in range(8))


def test_bench_spawns_its_own_ranks_when_launched_bare():
    """`python bench.py --gpus 2 ...` with no launcher (WORLD_SIZE unset), as the driver starts the N = 1 line: bench.py starts
    the ranks itself and still prints exactly one JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ZVX_BENCH_ENTRY"] = os.path.join(ROOT, "tests", "_bench_stub_main.py")      # the ranks run bench.main with the stub context
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_bench_stub_main.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "3", "--phonemes", "8", "--profile", "0", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["config"]["global_batch"] == 6 and r["output_ok"] is True
    assert r["config"]["wav_delivery"].startswith("device")


def test_comm_id_rendezvous_ignores_stray_connections(monkeypatch):
    """A connection that is not a rank (port scan, health probe: connects, sends nothing or junk) must not use up one of rank 0's
    world - 1 slots; a rank announces itself with magic + rank and is served once per distinct rank."""
    import socket, threading, time
    import bench
    probe = socket.socket(); probe.bind(("127.0.0.1", 0)); base = probe.getsockname()[1]; probe.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("ZVX_RDZV_PORT", str(base))
    cid = bytes(reversed(range(128)))
    got = {}
    srv = threading.Thread(target=lambda: got.__setitem__(0, bench.exchange_comm_id(0, 3, lambda: cid)))
    srv.start()
    time.sleep(0.3)
    for junk in (b"", b"GET / HTTP/1.0\r\n\r\n"):
        with socket.create_connection(("127.0.0.1", base), timeout=5) as c:
            if junk:
                c.sendall(junk)
            time.sleep(0.1)
    ths = [threading.Thread(target=lambda r=r: got.__setitem__(r, bench.exchange_comm_id(r, 3, lambda: b""))) for r in (1, 2)]
    for t in ths:
        t.start()
    for t in ths + [srv]:
        t.join(60)
    assert sorted(got) == [0, 1, 2] and all(got[r] == cid for r in got)


def test_traffic_profile_is_quoted_only_for_its_own_workload():
    """bench.workload_key: a PMC profile taken on V1 must not be quoted on a V3 line, nor config 2's on config 4."""
    import argparse
    import bench
    base = dict(config=2, decoder="styletts", vocoder="v1", precision="bf16", batch=None, phonemes=128)
    k = bench.workload_key(argparse.Namespace(**base))
    assert k["batch"] == 32 and k == bench.workload_key(argparse.Namespace(**dict(base, batch=32)))
    for change in (dict(vocoder="v3"), dict(config=4), dict(decoder="fastspeech2"), dict(precision="f32"), dict(batch=1), dict(config=5)):
        assert bench.workload_key(argparse.Namespace(**dict(base, **change))) != k
