#!/bin/bash
# round 5, GPU call H: conv-slab K-chunk prefetch (register-ring tiles <= 128 rows x 128 channels) -- tests + A/B against the
# previous build (zerovox_amd/libzvx_base.so, built from HEAD~ by hand; the script swaps the library file between runs)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5h; mkdir -p $OUT
cd $ROOT
L=zerovox_amd
timeout 1200 python -m pytest tests -m gpu -x -q -k "batch_flattened or decoders_alone or headline_utterance or e2e or padding_rows or front_end_under or fs2 or attention or encoder or refckpt or reference_written or speaker" > $OUT/pytest_sel.txt 2>&1; tail -4 $OUT/pytest_sel.txt
run() { # tag
  for i in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1_$1_$i.json 2>> $OUT/err.txt
  done
  timeout 300 python bench.py --no-cpu-baseline --steps 50 --set voc_f16=0 > $OUT/bench_n1bf_$1.json 2>> $OUT/err.txt
  timeout 300 python bench.py --no-cpu-baseline --steps 50 --decoder fastspeech2 > $OUT/bench_fs2_$1.json 2>> $OUT/err.txt
  timeout 300 python tools/enc_log.py 32 128 > $OUT/enc_$1.txt 2>> $OUT/err.txt
}
run new
cp $L/libzvx.so $L/libzvx_keep.so; cp $L/libzvx_base.so $L/libzvx.so
run base
cp $L/libzvx_keep.so $L/libzvx.so
run new2
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5h/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(j["ms_per_step"],3), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items()})
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/enc_new.txt $OUT/enc_base.txt
