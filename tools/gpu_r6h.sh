#!/bin/bash
# round 6, GPU call H: GPU suite (rb2fuse, conv2d_s2 with its last tap in LDS, half pre-norm sums of the FS2 decoder) + FS2 / config 5 A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6h; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt
for i in 1 2; do
timeout 300 python bench.py --decoder fastspeech2 --no-cpu-baseline > $OUT/bench_fs2_y16_$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --decoder fastspeech2 --set dec_y16=0 --no-cpu-baseline > $OUT/bench_fs2_y32_$i.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_cfg5.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6h/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), round(j["value"]), j.get("output_ok"), {k:round(v,3) for k,v in j.get("stage_ms_one_step_alone",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
grep -i "fs2\|fastspeech" $OUT/errlog.txt | head -20
tail -3 $OUT/bench.err
