"""Ad-hoc GPU parity sweep (development aid; the real checks live in tests/ -m gpu)."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zvx_oracle as O
from zerovox_amd import config as zcfg, weights as zw, pack, _lib

def err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max()), float(np.sqrt(np.mean((a - b) ** 2))), float(np.sqrt(np.mean(b ** 2)))

def make_ctx(kind, voc, prec):
    cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
    h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
    return _lib.Context(man, blob, 0), cfg, sd, h, hsd

def main():
    only = sys.argv[1:] 
    for prec in ("f32", "bf16"):
        for kind, voc in (("styletts", "tiny"), ("fastspeech2", "tiny2")):
            try:
                t0 = time.time()
                ctx, cfg, sd, h, hsd = make_ctx(kind, voc, prec)
                print(f"=== {prec} {kind} {voc} (create {time.time()-t0:.1f}s)", flush=True)
                r = np.random.default_rng(5)
                # vocoder alone
                mel = r.standard_normal((2, 12, 80)).astype(np.float32)
                P = np.array([12, 9], np.int32)
                wav = ctx.vocode_mel(mel, P)
                for b in range(2):
                    ref = O.hifigan_generator(mel[b, :P[b]].T, hsd, h)
                    print(f"  vocode_mel[{b}] max/rms/ref_rms", err(wav[b, :P[b] * 256], ref), flush=True)
                # decoder alone
                feats = r.standard_normal((2, 24, 528)).astype(np.float32)
                L = np.array([24, 17], np.int32)
                spk = r.standard_normal((2, 528)); spk = (spk / np.linalg.norm(spk, axis=1, keepdims=True)).astype(np.float32)
                melg = ctx.decode_features(feats, L, spk)
                for b in range(2):
                    ref = O.mel_decoder(feats[b, :L[b]], spk[b], sd, cfg)
                    print(f"  decode[{b}] max/rms/ref_rms", err(melg[b, :L[b]], ref), flush=True)
                # encoder
                T = np.array([16, 11], np.int32)
                ph = r.integers(0, 28, (2, 16)).astype(np.int32); pu = r.integers(0, 10, (2, 16)).astype(np.int32)
                mel_len, logd, pitch, energy = ctx.encode(ph, pu, T, spk)
                enc_out = ctx.fetch("encoder_out", (2, 16, 528))
                feats_g = ctx.fetch("features", (2, int(mel_len.max()), 528))
                for b in range(2):
                    ref = O.fs2_encoder(ph[b, :T[b]], pu[b, :T[b]], spk[b], sd, cfg)
                    print(f"  enc_out[{b}]", err(enc_out[b, :T[b]], ref["encoder_out"]), "logd", err(logd[b, :T[b]], ref["log_duration"])[0],
                          "pitch", err(pitch[b, :T[b]], ref["pitch"])[0], "energy", err(energy[b, :T[b]], ref["energy"])[0],
                          "mel_len", mel_len[b], ref["mel_len"], flush=True)
                    if mel_len[b] == ref["mel_len"]:
                        print(f"  features[{b}]", err(feats_g[b, :mel_len[b]], ref["features"]), flush=True)
                # full e2e
                out = ctx.synthesize(ph, pu, T, spk, duration=None, pad_to=np.array([40, 40], np.int32), Lmax_cap=400)
                for b in range(2):
                    ref = O.inference_ex(sd, hsd, cfg, h, ph[b, :T[b]], pu[b, :T[b]], spk[b], pad_to=40)
                    ml = ref["mel_len"]
                    print(f"  e2e[{b}] mel_len {out['mel_len'][b]} vs {ml}", "mel", err(out["mel"][b, :ml], ref["mel"].T) if out['mel_len'][b]==ml else None,
                          "wav", err(out["wav"][b, :ml * 256], ref["wav"]) if out['mel_len'][b]==ml else None, flush=True)
                # speaker encoder
                if kind == "styletts":
                    rm = r.standard_normal((2, 40, 80)).astype(np.float32); lens = np.array([40, 33], np.int32)
                    e = ctx.spkemb(rm, lens)
                    for b in range(2):
                        ref = O.resnet_se34v2(rm[b, :lens[b]], sd, cfg)
                        print(f"  spkemb[{b}]", err(e[b], ref), flush=True)
                ctx.close()
            except Exception:
                traceback.print_exc()

if __name__ == "__main__":
    main()
