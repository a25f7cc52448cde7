#!/bin/bash
# round 6, GPU call J: flipped-output direct epilogue on the ConvTranspose in front of the bf16 stage: V1 tests + headline + per-launch log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6j; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "v1 or ieee or config2 or config4 or pair_kernel or golden or audit or saturat or queued or history or invariance" > $OUT/pytest_sel.txt 2>&1; tail -4 $OUT/pytest_sel.txt
timeout 300 python tools/shape_log_b32.py > $OUT/shape_v1.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev_$i.json 2>> $OUT/bench.err; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6j/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), {k:round(v["ms_per_step"],3) for k,v in j.get("ab_vocoder_arithmetic",{}).items() if isinstance(v,dict)}, {r["stage"]:round(r["ms"],3) for r in j["roofline_per_stage"] if r["stage"].startswith("voc.up")})
    except Exception as e: print(f, "ERR", e)
PY
grep "N=1024 K=512\|K=256  taps=2\|K=128  taps=2\|N=64 " $OUT/shape_v1.txt | head
