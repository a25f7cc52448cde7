#!/bin/bash
# round 6, GPU call F: how much concurrency is left in the front end?  V2 / FS2 / V1 lines with 1, 2, 3 contexts in flight
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6f; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
for V in v2 v1; do for N in 1 2 3; do
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --vocoder $V --in-flight $N --no-cpu-baseline > $OUT/bench_${V}_inflight$N.json 2>> $OUT/bench.err
done; done
timeout 300 python bench.py --vocoder v2 --in-flight 2 --no-cpu-baseline > $OUT/bench_v2_inflight2_q4.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6f/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/bench.err
