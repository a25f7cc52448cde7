#!/bin/bash
# round 6, GPU call I: the full profile refresh on one box (bench lines, rocprofv3 kernel traces, PMC passes) + its `extra` half
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06a}
OUT=$ROOT/gpurun_out/r6i; mkdir -p $OUT
cd $ROOT
bash tools/refresh_profiles.sh $TAG > $OUT/refresh_$TAG.log 2>&1
bash tools/refresh_profiles.sh $TAG extra > $OUT/refresh_extra_$TAG.log 2>&1
ls $ROOT/gpurun_out/prof_$TAG | wc -l
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_*/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), round(j.get("roofline",{}).get("frac",0),3))
    except Exception as e: print(f, "ERR", e)
PY
