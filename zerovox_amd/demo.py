"""`python -m zerovox_amd.demo` -- the reference's `demo.py --iter N` benchmark loop (demo.py:99-138) on libzvx:
same RTF definition (audio seconds / synthesis seconds), same warm-up rule (iterations i <= 10 discarded), same prints.

    python -m zerovox_amd.demo --model synthetic:styletts --meldec-model synthetic:v1 --iter 30 "hello world, this is a test."
"""
import argparse
import time

import numpy as np

from .synthesize import ZeroVoxTTS, write_wav_to_file


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("text")
    ap.add_argument("--model", default="synthetic:styletts")
    ap.add_argument("--meldec-model", default="synthetic:v1")
    ap.add_argument("--infer-device", default="cuda")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--iter", type=int, default=1)
    ap.add_argument("--refmel-frames", type=int, default=258, help="synthetic 3 s reference mel for the speaker encoder")
    ap.add_argument("--wav-filename", default=None)
    args = ap.parse_args()

    modelcfg, synth = ZeroVoxTTS.load_model(args.model, args.meldec_model, infer_device=args.infer_device, precision=args.precision)
    sr = modelcfg["audio"]["sampling_rate"]
    print("computing speaker embedding...")
    refmel = np.random.default_rng(0).standard_normal((args.refmel_frames, modelcfg["audio"]["num_mels"])).astype(np.float32)
    spkemb = synth.speaker_embed_from_mel(refmel)
    rtf, warmup = [], 10
    for i in range(args.iter):
        t0 = time.time()
        wav, phoneme, length = synth.tts(args.text, spkemb)
        elapsed = time.time() - t0
        wav_len = wav.shape[0] / sr
        print(f"[{i + 1}/{args.iter}] Synth time: {elapsed:.2f} sec, voice length: {wav_len:.2f} sec, rtf: {wav_len / elapsed:.2f}")
        if args.wav_filename:
            write_wav_to_file(wav, length=length, filename=args.wav_filename, sample_rate=sr, hop_length=modelcfg["audio"]["hop_size"])
        if i > warmup:
            rtf.append(wav_len / elapsed)
    if rtf:
        print("Average RTF: {:.2f}".format(np.mean(rtf)))


if __name__ == "__main__":
    main()
