#!/bin/bash
# round 6, GPU call L: eight rows in flight in the InstanceNorm kernels: decoder tests + headline / V2 lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6l; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "decoder or golden or ragged or overlap or speaker or spk or tiny_utter or config2" > $OUT/pytest_sel.txt 2>&1; tail -3 $OUT/pytest_sel.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dev_$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2_$i.json 2>> $OUT/bench.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6l/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), {k:round(v,3) for k,v in j["stage_ms_one_step_alone"].items() if v}, [(r["stage"],r["ms"]) for r in j["roofline_per_stage"] if r["stage"]=="decoder.norm"])
    except Exception as e: print(f, "ERR", e)
PY
