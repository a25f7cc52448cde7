#!/bin/bash
# round 5, GPU call M: speaker encoder tests + config 5 A/B (slab_small bit 10 = 256-row tiles on the last level)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5m; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "speaker or spk or embed or resnet or refckpt or reference_written or determin" > $OUT/pytest_sel.txt 2>&1; tail -3 $OUT/pytest_sel.txt
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --config 5 > $OUT/bench_cfg5_new_$i.json 2>> $OUT/err.txt
  timeout 300 python bench.py --no-cpu-baseline --config 5 --set slab_small=2050 > $OUT/bench_cfg5_l2t128_$i.json 2>> $OUT/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5m/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(j["ms_per_step"],3), round(j["value"]), [(k["name"],k["launches"],k["ms"]) for k in j.get("kernels_one_step",[]) if "convslab" in k["name"]])
PY
