#!/bin/bash
# A/B of the conv-slab tile -> XCD remap (zvx_set_int "slab_flat": 0 per utterance, 1 over batch x tiles), same box, alternating;
# then a bit-equality check of a ragged synthesis call under both settings.   usage (on the GPU box): bash tools/ab_slab_flat.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
line() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_last_step']; print('%.3f ms/step  enc %.3f dec %.3f voc %.3f spk %.3f  ok=%s' % (d['ms_per_step'], s['encoder'], s['decoder'], s['vocoder'], s['spkemb'], d['output_ok']))"; }
for rep in 1 2; do
  for f in 0 1; do echo -n "headline     slab_flat=$f: "; line --set slab_flat=$f; done
done
for f in 0 1 0 1; do echo -n "fs2 decoder  slab_flat=$f: "; line --decoder fastspeech2 --set slab_flat=$f; done
for f in 0 1 0 1; do echo -n "config 5     slab_flat=$f: "; line --config 5 --steps 20 --set slab_flat=$f; done
for f in 0 1; do echo -n "config 4     slab_flat=$f: "; line --config 4 --set slab_flat=$f; done
for f in 0 1; do echo -n "B=1 T=64     slab_flat=$f: "; line --batch 1 --phonemes 64 --set slab_flat=$f; done
python - <<'P'
import numpy as np
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
for kind in ("styletts", "fastspeech2"):
    cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
    h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    for (B, T) in ((32, 128), (5, 70), (1, 64)):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 40, "const7")
        rng = np.random.default_rng(1)
        Tl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32); Tl[0] = T
        for b in range(B): ph[b, Tl[b]:] = 0; pu[b, Tl[b]:] = 0; dur[b, Tl[b]:] = 0
        pad = (dur.sum(1)).astype(np.int32)
        outs = []
        for f in (0, 1):
            ctx.set_int("slab_flat", f)
            r = ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=True)
            outs.append((r["mel"], r["wav"]))
        same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
        print(f"{kind} B={B} T={T} ragged: slab_flat 0 vs 1 bit-identical: {same}")
    mels = rng.standard_normal((250, 258, 80)).astype(np.float32); lens = rng.integers(129, 259, 250).astype(np.int32); lens[0] = 258
    e = []
    for f in (0, 1):
        ctx.set_int("slab_flat", f); e.append(ctx.spkemb(mels, lens))
    print(f"speaker encoder B=250 L=258: bit-identical: {np.array_equal(e[0], e[1])}")
    ctx.close()
P
