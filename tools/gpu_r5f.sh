#!/bin/bash
# round 5, GPU call F: whole GPU suite (error log) + the full profile refresh (one box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5f; mkdir -p $OUT
cd $ROOT
rm -f $OUT/errlog.txt
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
bash tools/refresh_profiles.sh r05k > $OUT/refresh.log 2>&1
bash tools/refresh_profiles.sh r05k extra > $OUT/refresh_extra.log 2>&1
ls $ROOT/gpurun_out/prof_r05k | wc -l
