#!/bin/bash
# round 6, GPU call N: the final profile refresh at the head sources (after the LayerNorm change): bench lines, kernel traces, PMC passes, power
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6n; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
bash tools/refresh_profiles.sh r06d > $OUT/refresh.log 2>&1
bash tools/refresh_profiles.sh r06d extra > $OUT/refresh_extra.log 2>&1
ls $ROOT/gpurun_out/prof_r06d | wc -l
