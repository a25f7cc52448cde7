#!/bin/bash
# round 5, GPU call C: vocoder in IEEE half -- whole GPU suite with the error log, headline A/B against bf16
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5c; mkdir -p $OUT
cd $ROOT
rm -f $OUT/errlog.txt
ZVX_ERR_LOG=$OUT/errlog.txt timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1_half_$i.json 2> $OUT/bench_n1.err
timeout 300 python bench.py --no-cpu-baseline --steps 50 --set voc_f16=0 > $OUT/bench_n1_bf16_$i.json 2>> $OUT/bench_n1.err
done
timeout 300 python bench.py --no-cpu-baseline --steps 50 --set dec_f16=0 > $OUT/bench_n1_decbf16.json 2>> $OUT/bench_n1.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --vocoder v2 > $OUT/bench_v2.json 2>> $OUT/bench_n1.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --vocoder v3 > $OUT/bench_v3.json 2>> $OUT/bench_n1.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --config 4 > $OUT/bench_cfg4.json 2>> $OUT/bench_n1.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5c/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline",{})
        print(os.path.basename(f), round(j["ms_per_step"],3), r.get("kernel"), round(r.get("frac",0),4), (r.get("alone") or {}).get("frac"), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items()})
        for s in j.get("roofline_per_stage",[])[:12]: print("     ", s["stage"], s["launches"], s["ms"], s["frac_mfma"])
    except Exception as e: print(f, "ERR", e)
PY
sort -t'|' -k2 $OUT/errlog.txt | awk -F'|' '{print $2, $3}' | sort | awk '{k=$1" "$2; } {print}' | sort -k1,2 | tail -60
