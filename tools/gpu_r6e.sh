#!/bin/bash
# round 6, GPU call E: GPU suite at the pruned sources (stage-2-bf16 default, two-unit gemm build) + bench lines + speaker-encoder sub-batching A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6e; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
for CH in 0 125 64 50 32; do
  timeout 300 python bench.py --no-cpu-baseline --config 5 --set spk_chunk=$CH > $OUT/bench_cfg5_chunk$CH.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --no-cpu-baseline --config 5 > $OUT/bench_cfg5_again.json 2>> $OUT/bench.err
timeout 300 python bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6e/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), round(j["value"]), j.get("output_ok"), {k:round(v["ms_per_step"],3) for k,v in j.get("ab_vocoder_arithmetic",{}).items() if isinstance(v,dict)})
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $OUT/bench.err
