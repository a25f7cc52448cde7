#!/bin/bash
# round 6, GPU call G: ResBlock2 in one launch (rb2fuse_kernel): V3 tests, V3 line A/B, per-launch log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6g; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "v3 or resblock2 or tiny2 or single_request_overlap or golden" > $OUT/pytest_sel.txt 2>&1; tail -5 $OUT/pytest_sel.txt
for i in 1 2; do
timeout 300 python bench.py --vocoder v3 --no-cpu-baseline > $OUT/bench_v3_fused_$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --vocoder v3 --set rb2fuse=0 --no-cpu-baseline > $OUT/bench_v3_unfused_$i.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --vocoder v3 --set front_overlap=0 --no-cpu-baseline > $OUT/bench_v3_fused_serial.json 2>> $OUT/bench.err
ZVX_VOCODER=v3 timeout 300 python tools/shape_log_b32.py > $OUT/shape_v3.txt 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6g/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), {k:round(v,3) for k,v in j.get("stage_ms_one_step_alone",{}).items()}, [(r["stage"],r["launches"],r["ms"]) for r in j.get("roofline_per_stage",[]) if r["stage"].startswith("voc.res")])
    except Exception as e: print(f, "ERR", e)
PY
grep "rb2fuse" $OUT/shape_v3.txt | head; tail -3 $OUT/bench.err
