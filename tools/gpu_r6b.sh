#!/bin/bash
# round 6, GPU call B: whole GPU suite at the new sources (async host delivery, saturation audit, narrow-stage isolation) + host-out A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6b; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev.json 2> $OUT/bench.err
timeout 300 python bench.py --host-out --no-cpu-baseline > $OUT/bench_host_async.json 2>> $OUT/bench.err
timeout 300 python bench.py --host-out --host-out-sync --no-cpu-baseline > $OUT/bench_host_sync.json 2>> $OUT/bench.err
timeout 300 python bench.py --host-out --pcm16 --no-cpu-baseline > $OUT/bench_host_async_pcm16.json 2>> $OUT/bench.err
timeout 600 python bench.py > $OUT/bench_n1.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6b/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), j.get("f32_mode"), j.get("busy_tail"))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $OUT/bench.err
