#!/bin/bash
# round 6, GPU call M: vectorised LayerNorm for every output form: whole GPU suite + per-launch log + headline / V2 lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6m; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python tools/shape_log_b32.py > $OUT/shape_v1.txt 2>&1
grep -n "helper.*encoder" $OUT/shape_v1.txt | head -12
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev_$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2_$i.json 2>> $OUT/bench.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6m/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), {k:round(v,3) for k,v in j["stage_ms_one_step_alone"].items() if v})
    except Exception as e: print(f, "ERR", e)
PY
