#!/bin/bash
# round 6, GPU call C: GPU suite again (audit test scale, host copies on the communication stream) + host-out A/B on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6c; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev_$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --host-out --no-cpu-baseline > $OUT/bench_host_async_$i.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --host-out --host-out-sync --no-cpu-baseline > $OUT/bench_host_sync.json 2>> $OUT/bench.err
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --host-out --no-cpu-baseline > $OUT/bench_host_async_q8.json 2>> $OUT/bench.err
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev_q8.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6c/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $OUT/bench.err
