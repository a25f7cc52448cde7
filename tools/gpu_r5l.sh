#!/bin/bash
# round 5, GPU call L: full GPU suite at the persistent-convolution sources + config 5 bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5l; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
grep -h "embed" $OUT/errlog.txt | tail -12
timeout 300 python bench.py --no-cpu-baseline --config 5 > $OUT/bench_cfg5.json 2>> $OUT/err.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2>> $OUT/err.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5l/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), round(j["value"]))
PY
