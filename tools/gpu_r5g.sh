#!/bin/bash
# round 5, GPU call G: 128-row tiles for badly quantised decoder launches -- A/B (slab_small bit 3 = off) + tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5g; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "batch_flattened or decoders_alone or headline_utterance or e2e or padding_rows or front_end_under" > $OUT/pytest_sel.txt 2>&1; tail -4 $OUT/pytest_sel.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1_t128_$i.json 2> $OUT/err.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 --set slab_small=10 > $OUT/bench_n1_off_$i.json 2>> $OUT/err.txt
done
timeout 300 python bench.py --no-cpu-baseline --steps 50 --decoder fastspeech2 > $OUT/bench_fs2_t128.json 2>> $OUT/err.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 --decoder fastspeech2 --set slab_small=10 > $OUT/bench_fs2_off.json 2>> $OUT/err.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5g/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        d=[s for s in j.get("roofline_per_stage",[]) if s["stage"] in ("decoder","decoder.norm")]
        print(os.path.basename(f), round(j["ms_per_step"],3), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items() if k=="decoder"}, [(s["stage"],s["launches"],s["ms"],s["frac_mfma"]) for s in d])
    except Exception as e: print(f, "ERR", e)
PY
