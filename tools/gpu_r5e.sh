#!/bin/bash
# round 5, GPU call E: narrow-stage kernel (V2) -- tests, A/B, trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5e; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "v2 or tiny or golden or saturates or many_short or streamed or single_request or batch_invariance" > $OUT/pytest_sel.txt 2>&1; tail -15 $OUT/pytest_sel.txt
timeout 300 python bench.py --no-cpu-baseline --steps 30 --vocoder v2 > $OUT/bench_v2.json 2> $OUT/bench_v2.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --vocoder v2 --set stagefuse=0 > $OUT/bench_v2_nofuse.json 2>> $OUT/bench_v2.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --vocoder v2 --set front_overlap=0 > $OUT/bench_v2_serial.json 2>> $OUT/bench_v2.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5e/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline",{})
        print(os.path.basename(f), round(j["ms_per_step"],3), r.get("kernel"), round(r.get("frac",0),4), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items()})
        for s in j.get("roofline_per_stage",[]):
            if s["stage"].startswith("voc."): print("     ", s["stage"], s["launches"], s["ms"], s["frac_mfma"], s["frac_hbm"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/bench_v2.err
