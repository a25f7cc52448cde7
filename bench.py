#!/usr/bin/env python3
"""bench.py -- headline benchmark: audio samples/s (+ RTF) for 128-phoneme zero-shot synthesis @22.05 kHz.

    python bench.py --gpus N --steps K --warmup W [--config 2|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

--config 2 (default, BASELINE.json configs[1]; N > 1 = configs[2]): a "step" = one pass of the whole hot path (phoneme
    encoder + variance adaptor + length regulator + mel decoder + HiFi-GAN) over 32 synthetic 128-phoneme utterances per
    GPU, forced durations = 7 -> 896 frames -> 229 376 samples (10.40 s) each, tts_medium_styledec + HiFi-GAN V1, seeded
    synthetic weights.  Weak scaling: every rank synthesises its own shard; each step ends with the ONE collective of the
    path, the waveform gather to rank 0 (RCCL send/recv issued inside libzvx, overlapped with the next step's synthesis).
--config 4 (configs[3]): HiFi-GAN V1 alone on device-resident 1024-frame N(0,1) mels (seed 7), --batch utterances per step.
--config 5 (configs[4]): ResNetSE34V2 speaker encoder on 1000 device-resident 3 s mels; a step = one batch of 250 clips.

Timing: W untimed steps, then exactly K steps bracketed by a barrier + full device drain on both sides; MAX over ranks.
Rank 0 prints ONE JSON line.  No GPU runtime other than libzvx (and, for N > 1, the system librccl it dlopens) is loaded and
torch is never imported; for N > 1 the 128-byte RCCL id travels over a plain TCP socket from rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK = {"bf16": 2500.0, "f32": 157.3}   # dense TFLOP/s (MI355X_MICROARCH.md)
F32_STAGES = ("variance",)                    # exact-f32 MFMA in both precision modes (its outputs are the discrete decisions)
SPLIT_STAGES = ("encoder",)                   # 16-bit mode: f32-class results from 3 half-precision products per multiply


def src_sha16():
    """Identity of the benched binary's sources: profiles/traffic.json must carry the same value to be quoted."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zerovox_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def workload_key(args):
    """What a PMC profile must have been taken on to be quoted for this run (profiles/traffic.json carries the same dict)."""
    return {"config": args.config, "decoder": args.decoder, "vocoder": args.vocoder, "precision": args.precision,
            "batch": args.batch or (250 if args.config == 5 else 32), "phonemes": args.phonemes,
            "exact_encoder": bool(getattr(args, "exact_encoder", False))}


def exchange_comm_id(rank, world, make_id):
    """rank 0's 128-byte RCCL id -> every rank over a plain TCP socket on MASTER_ADDR : (ZVX_RDZV_PORT or MASTER_PORT + 37).
    Deliberately NOT torch.distributed: importing torch after libzvx.so loads torch's bundled copies of librccl / libhsa-runtime
    next to the system HIP runtime libzvx is linked against, and RCCL then initialises against an HSA instance the process never
    opened (`pfn_hsa_system_get_info failed`, "no ROCm-capable device").  The bench process loads no GPU runtime but libzvx's."""
    import socket
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(os.environ.get("ZVX_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 37))
    ports = [base + 100 * i for i in range(4)]       # rank 0 binds the first free one; the others probe them in turn
    magic = b"ZVXID1"
    if rank == 0:
        cid = bytes(make_id())
        srv = None
        for port in ports:
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind(("127.0.0.1" if addr in ("127.0.0.1", "localhost") else addr, port))       # MASTER_ADDR only, never all interfaces
                break
            except OSError:
                srv.close(); srv = None
        if srv is None:
            raise SystemExit(f"rank 0: none of the rendezvous ports {ports} is free")
        srv.listen(world)
        srv.settimeout(600)
        served = set()                                 # distinct ranks that have the id: a stray connection (port scan, health
        while len(served) < world - 1:                  # probe) does not use up a slot, a retrying rank is served again
            conn, _ = srv.accept()
            with conn:
                try:
                    conn.settimeout(3)
                    hello = b""
                    while len(hello) < len(magic) + 4:
                        chunk = conn.recv(len(magic) + 4 - len(hello))
                        if not chunk:
                            break
                        hello += chunk
                    if len(hello) == len(magic) + 4 and hello.startswith(magic):
                        r = int.from_bytes(hello[len(magic):], "little")
                        if 0 < r < world:
                            conn.sendall(magic + cid)
                            served.add(r)
                except OSError:
                    pass
        srv.close()
        return cid
    t0, last, need = time.time(), None, len(magic) + 128
    while time.time() - t0 < 600:                  # rank 0 may still be packing weights: retry until it listens
        for port in ports:
            try:
                with socket.create_connection((addr, port), timeout=5) as c:
                    c.settimeout(3)                 # rank 0 answers at once; anything slower is not rank 0
                    c.sendall(magic + int(rank).to_bytes(4, "little"))
                    buf = b""
                    while len(buf) < need:
                        chunk = c.recv(need - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                    if len(buf) == need and buf.startswith(magic):
                        return buf[len(magic):]
            except OSError as e:
                last = e
        time.sleep(0.2)
    raise SystemExit(f"rank {rank}: no RCCL id from rank 0 at {addr}:{ports} ({last})")


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the environment
    torch.distributed.run would have set); rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    entry = os.environ.get("ZVX_BENCH_ENTRY", os.path.abspath(__file__))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, entry] + argv, env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


class SmiClockSampler:
    """Mean current gfx clock (all XCDs) from `amd-smi metric -c`, sampled from a side thread while the timed region runs: the
    clock the roofline's achieved figure was delivered at (the 2.5 PFLOP/s peak assumes 2.4 GHz).  Host-side only; absent tool or
    a region shorter than one sample -> no field."""

    def __init__(self, gpu):
        import threading
        self.gpu, self.samples, self.stop = gpu, [], False
        self.th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import subprocess
        while not self.stop:
            try:
                out = subprocess.run(["amd-smi", "metric", "-g", str(self.gpu), "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
                d = json.loads(out)
                d = d.get("gpu_data", d) if isinstance(d, dict) else d
                clk = d[0]["clock"]
                v = [c["clk"]["value"] for k, c in clk.items() if k.startswith("gfx_") and isinstance(c.get("clk"), dict) and isinstance(c["clk"].get("value"), (int, float))]
                if v and not self.stop:
                    self.samples.append(sum(v) / len(v))
            except Exception:
                return
            time.sleep(0.1)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.th.join(timeout=15)


def flush_c_stdio():
    import ctypes
    ctypes.CDLL(None).fflush(None)


def default_ctx_factory(args, local_rank):
    from zerovox_amd import _lib, config as zcfg, pack, weights as zw
    cfg = zcfg.medium_modelcfg(args.decoder)
    sd = zw.tts_state_dict(cfg, 0)
    hcfg = zcfg.hifigan_config(args.vocoder)
    hsd = zw.hifigan_state_dict(hcfg, 0)
    manifest, blob = pack.pack_model(cfg, sd, hcfg, hsd, args.precision)
    return _lib.Context(manifest, blob, local_rank), (cfg, sd, hcfg, hsd)


def cpu_baseline(config, model, T, pad_to, decoder="styletts", vocoder="v1", numpy_leg=False):
    """The oracle (a port of the reference's PyTorch CPU path) on a BOUNDED sample of the workload, timed on this box's host cores.
    Default: the oracle with its convolution primitives evaluated by torch / oneDNN in a child process (~10 s; the library the
    reference itself runs on a CPU).  numpy_leg (--cpu-numpy): also the plain NumPy / BLAS oracle (~30 s more), which is what
    stands alone when torch is missing on the box.  Checker / baseline only -- never on the product path."""
    cores = min(16, os.cpu_count() or 1)            # OpenBLAS / oneDNN stop scaling on these skinny conv GEMMs beyond ~16 threads
    unit = "clips/s" if config == 5 else "samples/s"
    res = _cpu_baseline_onednn(config, T, decoder, vocoder, cores, unit)
    if "value" in res and not numpy_leg:
        if config == 2:
            res.update(_reference_context())
        return res
    from oracle import zvx_oracle as O
    from zerovox_amd import synthetic
    from threadpoolctl import threadpool_limits
    cfg, sd, hcfg, hsd = model
    with threadpool_limits(limits=cores):
        O.hifigan_generator(np.zeros((80, 8), np.float32), hsd, hcfg)      # BLAS thread-pool warm-up
        t0 = time.time()
        if config == 2:
            ph, pu, spk, dur = synthetic.utterance(T, 0, "const7")
            out = O.inference_ex(sd, hsd, cfg, hcfg, ph, pu, spk, duration=dur, pad_to=pad_to)
            n, unit, what = len(out["wav"]), "samples/s", f"1 utterance of the workload ({T} phonemes -> {out['mel_len']} frames -> {len(out['wav'])} samples)"
        elif config == 4:
            mel = np.random.default_rng(7).standard_normal((80, 1024)).astype(np.float32)[:, :512]      # half an utterance keeps the sample within ~30 s
            n, unit, what = len(O.hifigan_generator(mel, hsd, hcfg)), "samples/s", "HALF an utterance of the workload: its first 512 of 1024 frames (-> 131072 samples)"
        else:
            mels = np.random.default_rng(8).standard_normal((4, 258, 80)).astype(np.float32)
            for m in mels:
                O.resnet_se34v2(m, sd, cfg)
            n, unit, what = 4, "clips/s", "4 clips of the workload (258-frame mels)"
        dt = time.time() - t0
    npres = {"value": n / dt, "sample": f"{what} through oracle/zvx_oracle.py (NumPy/BLAS fp32) in {dt:.1f} s"}
    if "value" in res:
        res["numpy_oracle"] = npres
    else:                                               # no torch on the box, or the child failed: the NumPy timing stands
        res = {"value": npres["value"], "unit": unit, "cores": int(cores), "kind": "port", "sample": npres["sample"], **res}
    if config == 2:
        res.update(_reference_context())
    return res


def _reference_context():
    # context, not measured here: the reference's own PyTorch / oneDNN CPU path in the survey container
    return {"reference_torch_cpu_samples_per_s": 189000,
            "reference_torch_cpu_note": "gooofy/zerovox inference_ex on 8 cores of the survey container (BASELINE.md section 3); the oneDNN-backed port is the comparable figure, the plain NumPy oracle (--cpu-numpy) is ~12x slower than that path"}


def _cpu_baseline_onednn(config, T, decoder, vocoder, cores, unit):
    """The oracle with its three convolution primitives evaluated by torch / oneDNN -- the library the reference itself runs on a CPU --
    in a child process (this process holds the HIP runtime of libzvx; torch brings its own).  The child first checks the port against
    the plain NumPy oracle on a small case."""
    import subprocess
    units = {2: 8, 4: 6, 5: 80}[config]                   # ~10 s of CPU work on 16 threads
    try:
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "onednn_port.py"), str(config),
                              decoder, vocoder, str(cores), str(units), str(T)],
                             capture_output=True, text=True, timeout=600, env={**os.environ, "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
        j = json.loads(out.stdout.strip().splitlines()[-1])
        return {"value": j["value"], "unit": unit, "cores": int(cores), "kind": "port",
                "sample": (f"{j['what']} through oracle/zvx_oracle.py with conv1d / conv_transpose1d / conv2d evaluated by torch {j['torch']} "
                           f"(oneDNN, {j['threads']} threads; checked against the NumPy oracle first: max |diff| {j['check_max_abs_diff']:.1e}) "
                           f"in {j['seconds']:.1f} s")}
    except Exception as e:
        return {"onednn_port_error": f"{type(e).__name__}: {e}"[:300]}


def main(argv=None, ctx_factory=default_ctx_factory):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5])
    ap.add_argument("--batch", type=int, default=None, help="units per GPU and step (default: 32 utterances / 250 clips)")
    ap.add_argument("--phonemes", type=int, default=128)
    ap.add_argument("--decoder", default="styletts", choices=["styletts", "fastspeech2"])
    ap.add_argument("--vocoder", default="v1")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--pcm16", action="store_true", help="int16 PCM waveform rows (halves the gather payload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-numpy", action="store_true", help="cpu_baseline: also time the plain NumPy / BLAS oracle (~30 s more; it always runs when the oneDNN-backed port cannot)")
    ap.add_argument("--exact-encoder", action="store_true", help="bf16 mode: phoneme encoder on the exact-f32 MFMA instead of 3-plane split products on IEEE-half planes (A/B: both are f32-class)")
    ap.add_argument("--host-out", action="store_true", help="config 2: deliver every step's waveform to host memory inside the timed region, as the reference's tts_ex does (asynchronously: pinned slots + copy stream, the host takes step i's rows while step i+1 runs)")
    ap.add_argument("--host-out-sync", action="store_true", help="with --host-out: the round-5 form, every call waits for its own D2H copy (A/B)")
    ap.add_argument("--in-flight", type=int, default=1, help="config 2, one GPU: steps alternate over this many contexts (A/B of round 3; since round 4 ONE context overlaps batch i+1's front end with batch i's vocoder by itself: zvx_set_int front_overlap)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=INT", help="zvx_set_int(KEY, INT) on every context before the first step (A/B of a runtime switch; echoed in config.overrides)")
    ap.add_argument("--profile", type=int, default=2, help="0 none, 1 stage events, 2 + per-launch events on the dominant kernel")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 and args.config != 2:
        raise SystemExit("configs 4 / 5 are single-GPU isolation benchmarks")
    if args.host_out and (world > 1 or args.config != 2):
        raise SystemExit("--host-out is the single-GPU config-2 variant")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from zerovox_amd import synthetic
    ctx, model = ctx_factory(args, local_rank)       # raises when libzvx.so / a GPU is missing: there is no CPU fallback
    cfg = model[0]
    hop, sr = cfg["audio"]["hop_size"], cfg["audio"]["sampling_rate"]
    if args.exact_encoder:
        ctx.set_int("enc_split", 0)
    overrides = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.set}
    for k, v in overrides.items(): ctx.set_int(k, v)
    if world > 1:
        ctx.comm_init(exchange_comm_id(rank, world, ctx.comm_unique_id), rank, world)
        flush_c_stdio()                              # RCCL's version banner (C stdio, block-buffered on a pipe) goes out now, not after the JSON line
    else:
        ctx.comm_init(None, 0, 1)

    rccl_info = None
    if world > 1 and hasattr(ctx, "comm_info"):
        try:
            rccl_info = ctx.comm_info()              # collective: every rank calls it
        except Exception as e:
            rccl_info = {"error": f"{type(e).__name__}: {e}"[:200]}
    T = args.phonemes
    more_ctx = []                                    # --in-flight: (context, [wav buffers]) beyond the first
    ss = 2 if args.pcm16 else 4
    wdt = np.int16 if args.pcm16 else np.float32
    if args.config == 2:
        B = args.batch or 32
        ph, pu, Tlen, spk, dur = synthetic.batch(B, T, first_utt=rank * B, dur_mode="const7")
        L = int(dur[0].sum()); N = L * hop
        pad_to = np.full(B, max(689, L), np.int32)          # fresh-model semantics of model.py:331-335
        row_bytes = N * ss
        wav = [ctx.dev_alloc(B * row_bytes) for _ in range(2)]                       # double-buffered: the gather of step i overlaps step i + 1
        gathered = ctx.dev_alloc(world * B * row_bytes) if (world > 1 and rank == 0) else 0
        units_per_step, unit = B * N, "samples/s"
        it = [0]

        host_wav = [None]
        host_prev = [None]                               # --host-out: slot of the step whose rows the host has not taken yet

        # --in-flight n (one GPU): n - 1 further contexts, each with its own stream and work buffers; steps go round robin, every step
        # is still one whole batch and all of them are complete at the closing fence
        if args.in_flight > 1 and (world > 1 or args.host_out):
            raise SystemExit("--in-flight > 1 is a one-GPU, device-output mode")
        for _ in range(max(1, args.in_flight) - 1):
            c2, _m = ctx_factory(args, local_rank)
            if args.exact_encoder:
                c2.set_int("enc_split", 0)
            for k, v in overrides.items(): c2.set_int(k, v)
            c2.comm_init(None, 0, 1)
            more_ctx.append((c2, [c2.dev_alloc(B * row_bytes) for _ in range(2)]))

        def step():
            if args.host_out and args.host_out_sync:    # the reference's contract: tts_ex returns host NumPy (synthesize.py:233-239)
                host_wav[0] = ctx.synthesize(ph, pu, Tlen, spk, dur, pad_to, want_mel=False, pcm16=args.pcm16)["wav"]
                return
            if args.host_out:
                # same contract, pipelined: step i is queued (ZVX_HOST_ASYNC: its rows travel to a pinned host slot on the context's
                # copy stream), then the host takes delivery of step i - 1's rows -- every step's waveform is in host memory when
                # the closing fence returns
                r = ctx.synthesize(ph, pu, Tlen, spk, dur, pad_to, want_mel=False, pcm16=args.pcm16, host_async=True)
                if host_prev[0] is not None:
                    host_wav[0] = ctx.wait_host(host_prev[0], pcm16=args.pcm16)
                host_prev[0] = r["slot"]
                return
            i = it[0]; it[0] += 1
            nf = 1 + len(more_ctx)
            c, bufs = (ctx, wav) if i % nf == 0 else more_ctx[i % nf - 1]
            buf = bufs[(i // nf) & 1]
            c.synthesize(ph, pu, Tlen, spk, dur, pad_to, want_mel=False, wav_device_ptr=buf, wav_stride=N, no_sync=True, pcm16=args.pcm16)
            if world > 1:
                ctx.comm_gather(buf, B * row_bytes, gathered, root=0, no_sync=True)
        workload = (f"batch={B}/GPU x {T}-phoneme utterances, precomputed spk-embed, forced durations=7 -> {L} frames -> {N} samples each; "
                    f"tts_medium_{'styledec' if args.decoder == 'styletts' else 'fs2'} + HiFi-GAN {args.vocoder.upper()}, end-to-end phoneme->waveform"
                    + (f", RCCL waveform gather ({'int16' if args.pcm16 else 'f32'}) to rank 0 each step" if world > 1 else ""))
        cfg_extra = {"global_batch": B * world, "phonemes": T, "frames": L, "samples_per_utt": N, "decoder": args.decoder,
                     "vocoder": args.vocoder, "pad_to": int(pad_to[0]), "wav_dtype": "int16" if args.pcm16 else "f32",
                     "encoder_arithmetic": ("exact f32 MFMA" if (args.exact_encoder or args.precision == "f32" or overrides.get("enc_split") == 0) else
                                            ("3-plane split products on bf16 planes (rounds 2-3 A/B: 5e-5 on the encoder output)" if overrides.get("enc_split") == 1 else
                                             "3-plane split products on IEEE-half planes (hi.wh + hi.wl + lo.wh, every operand carried to 2^-24: f32-class, discrete decisions held to the f32 mode's 1e-3 margin)")),
                     "in_flight": max(1, args.in_flight),
                     "front_overlap": ("off" if overrides.get("front_overlap") == 0 else
                                       "one context, two streams: the encoder / variance adaptor / mel decoder of step i+1 are queued on the context's front stream and run under the vocoder of step i (every step is a whole batch, all complete at the closing fence; bit-identical to the serial schedule)"),
                     "decoder_arithmetic": ("f32" if args.precision == "f32" else
                                            "IEEE half weights + activations on the f16 MFMA (log-mel within 2e-2 of the f32 reference)"),
                     "vocoder_arithmetic": ("f32" if args.precision == "f32" else
                                            ("bf16 weights + activations + running sum (rounds 1-4; A/B: --set voc_f16=0)" if overrides.get("voc_f16", 1) == 0 else
                                             ("per-stage mask --set voc_f16_stages (bit k: domain k in IEEE half, else bf16)" if "voc_f16_stages" in overrides else
                                              "IEEE half weights + activations + running sum (saturating stores) everywhere except the 128-channel ResBlock1 stage, which runs in bf16 "
                                              "(round 6: the f16 multiplier array's extra power costs that stage 0.3 ms of the 0.8-1.0 ms an all-half generator costs; ab_vocoder_arithmetic has all three)"))),
                     "wav_delivery": ("host (synchronous D2H copy of every step's waveform inside the timed region: every call waits)" if (args.host_out and args.host_out_sync) else
                                      "host (ZVX_HOST_ASYNC: every step's waveform lands in a pinned host slot via the context's copy stream inside the timed region; "
                                      "the host takes step i's rows while step i+1 runs)") if args.host_out else
                                     "device (rows stay in HBM for the gather / the caller; --host-out times the D2H copy too)"}
    elif args.config == 4:
        B = args.batch or 32
        P = 1024; N = P * hop
        mel = np.random.default_rng(7).standard_normal((B, P, 80)).astype(np.float32)
        mel_d = ctx.dev_alloc(mel.nbytes); ctx.dev_from_host(mel_d, mel)
        wav = [ctx.dev_alloc(B * N * ss)]
        Pn = np.full(B, P, np.int32)
        units_per_step, unit = B * N, "samples/s"

        def step():
            ctx.vocode_mel_device(mel_d, Pn, P, wav[0], N, no_sync=True, pcm16=args.pcm16)
        workload = f"HiFi-GAN {args.vocoder.upper()} generator alone: batch={B} x 1024-frame N(0,1) mels (seed 7, device-resident) -> {N} samples each"
        cfg_extra = {"global_batch": B, "frames": P, "samples_per_utt": N, "vocoder": args.vocoder}
    else:
        B = args.batch or 250
        Tr = 258
        mels = np.random.default_rng(8).standard_normal((B, Tr, 80)).astype(np.float32)
        mel_d = ctx.dev_alloc(mels.nbytes); ctx.dev_from_host(mel_d, mels)
        emb_d = ctx.dev_alloc(B * ctx.hidden * 4)
        lens = np.full(B, Tr, np.int32)
        units_per_step, unit = B, "clips/s"

        def step():
            ctx.spkemb_device(mel_d, lens, B, Tr, emb_d, no_sync=True)
        workload = (f"ResNetSE34V2 speaker encoder: batches of {B} device-resident 3 s reference mels [258, 80] "
                    f"({args.steps} steps = {args.steps * B} clips; BASELINE configs[4] = 1000 clips = 4 steps of 250)")
        cfg_extra = {"global_batch": B, "ref_frames": Tr}

    def fence():
        for c2, _b in more_ctx:
            c2.comm_barrier()
        ctx.comm_barrier()                            # drains every stream of every rank (the copy stream too), then all ranks arrive (world 1: just the drain)
        if args.config == 2 and args.host_out and not args.host_out_sync and host_prev[0] is not None:
            host_wav[0] = ctx.wait_host(host_prev[0], pcm16=args.pcm16)      # the last queued step's rows (already landed: the fence drained the copy stream)

    for _ in range(args.warmup):
        step()
    # One extra untimed step with events on every launch gives the per-stage roofline split and finds the dominant kernel
    # variant; inside the timed region only that variant's launches carry events (each timed launch costs ~4 us).
    kstats_all, tstats, stage_ms_alone = [], [], None
    if args.profile >= 2:
        fence()
        ctx.set_int("profile", 2)
        ctx.set_int("profile_only", -1)
        ctx.reset_stats()
        if more_ctx:
            it[0] += (-it[0]) % (1 + len(more_ctx))   # --in-flight: the profiled step must be one of the first context's
        step()
        fence()
        kstats_all, tstats = ctx.kernel_stats(), ctx.tag_stats()
        stage_ms_alone = ctx.stage_times()            # this step ran alone on the chip (nothing queued behind it)
        if kstats_all:
            dom_name = max(kstats_all, key=lambda k: k["ms"])["name"]
            ctx.set_int("profile_only", ctx.get_int("variant_id:" + dom_name))
    ctx.set_int("profile", args.profile)
    ctx.reset_stats()
    fence()
    smi = SmiClockSampler(local_rank) if rank == 0 else None
    if smi:
        smi.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if smi:
        smi.__exit__()
    stage_ms = ctx.stage_times()
    kstats = ctx.kernel_stats()
    ctx.set_int("profile", 0)
    # A/B inside the same process, on the same box, minutes... seconds apart: the vocoder's arithmetic.  The product default since round 5 is
    # IEEE half (8x less rounding noise in the waveform); the bf16 kernels of rounds 1-4 stay behind zvx_set_int("voc_f16", 0) and are
    # ~4-5 % faster at the board's power limit.  Untimed for the line's `value`; reported beside it so that one run carries both figures.
    voc_ab = None
    if world == 1 and args.config in (2, 4) and args.precision == "bf16" and "voc_f16" not in overrides and "voc_f16_stages" not in overrides \
            and not args.host_out and args.in_flight <= 1 and not os.environ.get("ZVX_BENCH_NO_AB") and args.steps >= 10:
        # measured error of each choice at the headline shape (tools/ab_voc_stages.py, profiles/r06_ab_voc_stages.txt): waveform error against
        # the f32 oracle end to end / the generator alone, max | rms
        notes = {"all_bf16": ("voc_f16", 0, "bf16 everywhere (rounds 1-4): 7.3e-3 | 1.36e-3 end to end, 6.7e-3 | 1.33e-3 alone"),
                 "all_half": ("voc_f16_stages", 31, "IEEE half everywhere (round 5's default): 1.64e-3 | 3.1e-4 end to end, 1.02e-3 | 1.7e-4 alone")}
        voc_ab = {"default": "IEEE half everywhere except the 128-channel ResBlock1 stage (the pair kernel: 43 % of the generator's matrix work) in bf16: "
                             "3.55e-3 | 6.6e-4 end to end, 3.33e-3 | 6.1e-4 alone (SURVEY 8c allows 1e-2 | 2e-3)"}
        try:
            for name, (key, val, note) in notes.items():
                ctx.set_int(key, val)
                for _ in range(3):
                    step()
                fence()
                n_ab = min(args.steps, 40)
                t1 = time.perf_counter()
                for _ in range(n_ab):
                    step()
                fence()
                dt_ab = time.perf_counter() - t1
                voc_ab[name] = {"set": f"{key}={val}", "steps": n_ab, "ms_per_step": 1e3 * dt_ab / n_ab, "value": units_per_step * n_ab / dt_ab, "wav_err_max_rms": note}
                ctx.set_int("voc_f16", 1); ctx.set_int("voc_f16_stages", -1)
        finally:
            ctx.set_int("voc_f16", 1); ctx.set_int("voc_f16_stages", -1)
            step(); fence()                           # (back on the default kernels before anything else runs)
    # a timed region shorter than an external sampler's period (amd-smi at 1-5 s) would read as "GPU idle": keep the chip busy for >= 6 s
    # more, OUTSIDE the timed region and before the CPU baseline (VERDICT r4 #8d, r5 #6b: 2 s was still shorter than a 5 s sampler); the
    # tail is itself timed and reported (`busy_tail`: a second, longer measurement of the same steps on the same box), counted in nothing
    busy_steps, busy_tail = 0, None
    if world == 1 and elapsed < 6.0 and not args.no_cpu_baseline and not os.environ.get("ZVX_BENCH_NO_BUSY_TAIL"):   # (the default command; profiling passes use --no-cpu-baseline)
        per = max(elapsed / max(args.steps, 1), 1e-4)
        busy_steps = int(min(6.5 / per, 20000))
        tb = time.perf_counter()
        for _ in range(busy_steps):
            step()
        fence()
        tb = time.perf_counter() - tb
        busy_tail = {"steps": busy_steps, "seconds": tb, "ms_per_step": 1e3 * tb / max(busy_steps, 1),
                     "note": "untimed for `value`: the same step() right behind the timed region, long enough for a 5 s utilisation sampler to see the GPU busy"}
    # The reference computes in fp32 everywhere (synthesize.py:228-233, no autocast): the same workload once more in the library's
    # exact-f32 mode (every contraction on v_mfma_f32_32x32x2_f32, the mode every parity test holds to 2e-4) -- a few steps on a second
    # context, untimed for `value`, quoted beside it as `f32_mode` (VERDICT r5 #6a).  The default command only.
    f32_mode = None
    if world == 1 and args.config == 2 and args.precision == "bf16" and not args.no_cpu_baseline and not overrides and not args.host_out \
            and args.in_flight <= 1 and not os.environ.get("ZVX_BENCH_NO_F32"):
        try:
            import copy
            a32 = copy.copy(args); a32.precision = "f32"
            c32, _m32 = ctx_factory(a32, local_rank)
            c32.comm_init(None, 0, 1)
            w32 = c32.dev_alloc(B * row_bytes)
            def step32():
                c32.synthesize(ph, pu, Tlen, spk, dur, pad_to, want_mel=False, wav_device_ptr=w32, wav_stride=N, no_sync=True, pcm16=args.pcm16)
            step32(); c32.comm_barrier()
            n32 = 4
            t32 = time.perf_counter()
            for _ in range(n32):
                step32()
            c32.comm_barrier()
            t32 = time.perf_counter() - t32
            f32_mode = {"dtype": "f32", "steps": n32, "ms_per_step": 1e3 * t32 / n32, "value": units_per_step * n32 / t32, "unit": unit,
                        "note": "the reference's arithmetic (fp32 everywhere) on the exact-f32 MFMA: same workload, same box, a second context, untimed for `value`"}
            c32.dev_free(w32); c32.close()
        except Exception as e:
            f32_mode = {"error": f"{type(e).__name__}: {e}"[:200]}
    elapsed = ctx.comm_max(elapsed)                  # MAX over ranks

    # sanity on the produced data (not timed)
    if args.config == 5:
        e = ctx.dev_to_host(emb_d, (B, ctx.hidden), np.float32)
        ok = bool(np.isfinite(e).all()) and float(np.abs(np.linalg.norm(e, axis=1) - 1).max()) < 1e-3
    else:
        if world > 1 and rank == 0:
            g = ctx.dev_to_host(gathered, (world * B, N), wdt)
            own = ctx.dev_to_host(wav[(it[0] - 1) & 1], (B, N), wdt)           # (world > 1 runs one context: it[0] counts its steps)
            ok = bool(np.array_equal(g[:B], own)) and all(bool(np.abs(g[r * B:(r + 1) * B].astype(np.float32)).max() > 0) for r in range(world))
        else:
            ok = True
        w = (host_wav[0] if (args.config == 2 and args.host_out) else ctx.dev_to_host(wav[0], (B, N), wdt))[0].astype(np.float32) / (32760.0 if args.pcm16 else 1.0)
        ok = ok and bool(np.isfinite(w).all()) and 0 < float(np.abs(w).max()) <= 1.0

    # the arithmetic type of the workload's dominant stage: the 16-bit mode ("--precision bf16", the manifest's name for it) computes the
    # vocoder and both mel decoders on the IEEE-half MFMA since round 5 (zvx_set_int voc_f16 / dec_f16 0: bf16), the speaker encoder in bf16
    if args.precision == "f32":
        dtype_name = "f32"
    elif args.config == 5:
        dtype_name = "bf16"
    else:
        # the generator's arithmetic is chosen per stage (zvx_set_int voc_f16_stages): by default IEEE half everywhere except V1's 128-channel
        # stage, whose pair kernel -- the step's dominant kernel -- runs in bf16; both are 16-bit MFMA types of the same rate
        mask = overrides.get("voc_f16_stages", -1)
        mixed = args.vocoder == "v1" and mask < 0
        dtype_name = "bf16" if overrides.get("voc_f16", 1) == 0 or mask == 0 else ("f16+bf16" if (mixed or (0 < mask < 31 and mask & 31 != 31)) else "f16")
    if rank == 0:
        total = float(world) * units_per_step * args.steps
        value = total / elapsed
        res = {
            "metric": "audio samples/sec + RTF, 128-phoneme zero-shot synth @22.05kHz, 1/2/4/8 GPU" if args.config == 2 else
                      ("audio samples/sec, HiFi-GAN generator alone (BASELINE configs[3])" if args.config == 4 else
                       "speaker embeddings/sec, ResNetSE34V2 on 3 s reference mels (BASELINE configs[4])"),
            "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "timed_region_s": elapsed, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": workload, **cfg_extra, **({"overrides": overrides} if overrides else {})},
            "stage_ms_last_step": stage_ms, "output_ok": ok, "src_sha16": src_sha16(), "untimed_busy_tail_steps": busy_steps,
            "steps_for_2s": int(np.ceil(2.0 / max(elapsed / max(args.steps, 1), 1e-6))),      # --steps that would make the timed region 2 s on this box
        }
        if busy_tail is not None:
            res["busy_tail"] = busy_tail
        if f32_mode is not None:
            res["f32_mode"] = f32_mode
        if voc_ab is not None:
            res["ab_vocoder_arithmetic"] = voc_ab         # same process, same box, right behind the timed region: the two uniform choices beside the default
            if "all_bf16" in voc_ab:
                res["ab_voc_bf16"] = voc_ab["all_bf16"]   # (the round-5 key, kept for readers of earlier lines)
        if stage_ms_alone is not None:
            # stage_ms_last_step are event pairs on the stream a stage runs on: with the front end of step i+1 queued under the vocoder of
            # step i (config.front_overlap) the encoder / decoder figures are WALL times of work that waits for free CUs most of the time;
            # the instrumented step below ran alone and gives the stages' own durations
            res["stage_ms_one_step_alone"] = stage_ms_alone
        if rccl_info is not None:
            res["rccl"] = rccl_info                   # ranks the communicator itself counts, RCCL version, every rank's device
        if unit == "samples/s":
            audio_s = total / sr
            res["rtf_ref_audio_s_per_s"] = audio_s / elapsed
            res["rtf_s_per_audio_s"] = elapsed / audio_s
        if kstats:
            dom = max(kstats, key=lambda k: k["ms"])
            avg_ms = dom["ms"] / dom["launches"]
            achieved = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
            f32_dom = dom["name"].startswith("gemm_f32")
            peak = MFMA_PEAK["f32" if (args.precision == "f32" or f32_dom) else "bf16"]
            traffic = traffic_note = None
            pmc = {}
            # per-launch HBM bytes of this kernel from the rocprofv3 PMC passes of the SAME sources on the SAME workload
            # (tools/refresh_profiles.sh writes one profiles/traffic*.json per profiled command)
            import glob
            seen = []
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic*.json"))):
                try:
                    tj = json.load(open(f))
                except Exception:
                    continue
                seen.append((os.path.basename(f), tj.get("src_sha16"), tj.get("workload_key")))
                if tj.get("src_sha16") == res["src_sha16"] and tj.get("workload_key") == workload_key(args):
                    pmc = tj.get(dom["name"], {})
                    traffic = pmc.get("hbm_bytes_per_launch")
                    res["traffic_source"] = "profiles/" + os.path.basename(f)
                    break
            else:
                traffic_note = ("no profiles/traffic*.json was collected on these sources AND this workload: not quoted "
                                f"(found {[(n, sh) for n, sh, _ in seen]})") if seen else "profiles/traffic*.json missing"
            res["roofline"] = {"bound": "mfma", "kernel": dom["name"], "achieved": achieved, "peak": peak,
                               "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                               "launches": dom["launches"], "avg_launch_ms": avg_ms,
                               "flops_per_launch": dom["flops"] / dom["launches"],
                               "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
                               "alg_GBps": dom["bytes"] / (dom["ms"] * 1e-3) / 1e9,
                               "measured": "per-launch HIP events (carried by the dispatches) of every launch of this kernel inside the timed region"}
            ser = next((k for k in kstats_all if k["name"] == dom["name"] and k["launches"]), None)
            overlapped = args.config == 2 and not args.host_out and overrides.get("front_overlap", 1) != 0
            if ser and overlapped:
                # Queued steps overlap: while this kernel runs, the NEXT step's encoder / decoder launches share the chip with it, so a
                # launch in the timed region can last longer than the kernel needs alone.  The top-level achieved / frac are the TIMED
                # REGION's (what the contract asks for, what rocprofv3 --kernel-trace of this command sees, and what goes with ms_per_step);
                # `alone` holds the same kernel's launches in the one untimed, fully instrumented step, which ran alone on the chip
                # (rocprofv3 of `bench.py --set front_overlap=0` agrees with that one).
                ser_tf = ser["flops"] / (ser["ms"] * 1e-3) / 1e12
                res["roofline"]["alone"] = {"achieved": ser_tf, "frac": ser_tf / peak, "launches": ser["launches"], "avg_launch_ms": ser["ms"] / ser["launches"],
                                            "alg_GBps": ser["bytes"] / (ser["ms"] * 1e-3) / 1e9,
                                            "measured": "per-launch HIP events of this kernel's launches in the ONE untimed, fully instrumented step of this run "
                                                        "(nothing queued behind it: no other step's front end on the chip)"}
                res["roofline"]["frac_timed_region"] = res["roofline"]["frac"]
                res["roofline"]["frac_alone"] = ser_tf / peak
            if args.in_flight > 1:
                res["roofline"]["note"] = ("launch durations measured while the other context's launches share the CUs (two vocoders side by side "
                                           "stretch each launch): the kernel's own roofline is on the in_flight = 1 line")
            if traffic_note:
                res["roofline"]["traffic_note"] = traffic_note
            if smi and smi.samples:
                res["roofline"]["smi_gfx_clock_GHz"] = round(sum(smi.samples) / len(smi.samples) / 1e3, 3)      # amd-smi, during the timed region
                res["roofline"]["smi_clock_samples"] = len(smi.samples)
            if pmc.get("eff_clock_GHz") and pmc["eff_clock_GHz"] <= 2.4:   # GRBM_GUI_ACTIVE / duration over-counts on sub-100-us launches: only a physical value is quoted
                # the same profile's GRBM_GUI_ACTIVE / kernel time and MFMA busy cycles: `peak` above is the 2.4 GHz figure, the
                # chip sustains less under this load (power budget), and `mfma_busy_frac` is the share of THOSE cycles the pipe works
                res["roofline"]["profiled_clock_GHz"] = round(pmc["eff_clock_GHz"], 3)
                res["roofline"]["profiled_mfma_busy_frac"] = round(pmc.get("mfma_busy_frac", 0.0), 3)
            # per-variant and per-stage split of ONE (untimed, fully instrumented) step
            res["kernels_one_step"] = [{"name": k["name"], "launches": k["launches"], "ms": round(k["ms"], 3),
                                        "TFLOPs": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2) if k["ms"] > 0 else None}
                                       for k in sorted(kstats_all, key=lambda k: -k["ms"])]
            per_stage = []
            for t in sorted(tstats, key=lambda t: -t["ms"]):
                if t["ms"] <= 0:
                    continue
                pk = MFMA_PEAK["f32" if (args.precision == "f32" or t["name"] in F32_STAGES) else "bf16"]
                tf, gb = t["flops"] / (t["ms"] * 1e-3) / 1e12, t["bytes"] / (t["ms"] * 1e-3) / 1e9
                row = {"stage": t["name"], "launches": t["launches"], "ms": round(t["ms"], 4), "TFLOPs": round(tf, 2),
                       "alg_GBps": round(gb, 1), "frac_mfma": round(tf / pk, 4), "frac_hbm": round(gb / HBM_PEAK_GBS, 4), "mfma_peak": pk}
                if args.precision == "bf16" and t["name"] in SPLIT_STAGES:
                    row["note"] = "algorithmic (f32-equivalent) FLOPs; issued on the f16 MFMA as ~3x that (hi.wh + hi.wl + lo.wh)"
                per_stage.append(row)
            res["roofline_per_stage"] = per_stage
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.config, model, T, 896 if args.config == 2 else 0, args.decoder, args.vocoder, numpy_leg=args.cpu_numpy)
        flush_c_stdio()
        print(json.dumps(res), flush=True)
    fence()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
