#!/bin/bash
# round 6, GPU call P: head sources (C = 64 fused-pool instantiation removed): whole GPU suite, smoke, then the final profile refresh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6p; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
bash tools/refresh_profiles.sh r06e > $OUT/refresh.log 2>&1
bash tools/refresh_profiles.sh r06e extra > $OUT/refresh_extra.log 2>&1
ls $ROOT/gpurun_out/prof_r06e | wc -l
