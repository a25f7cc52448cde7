#!/bin/bash
# round 6, GPU call K: final validation at the head sources: whole GPU suite, smoke, race screens (default = mixed generator arithmetic, V3 fused ResBlock2), then the profile refresh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6k; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 900 python tools/race_hunt.py 120 > $OUT/race_hunt.txt 2>&1; cat $OUT/race_hunt.txt | cut -c1-200
timeout 900 python tools/race_hunt_all.py > $OUT/race_hunt_all.txt 2>&1; tail -12 $OUT/race_hunt_all.txt | cut -c1-200
bash tools/refresh_profiles.sh r06b > $OUT/refresh.log 2>&1
bash tools/refresh_profiles.sh r06b extra > $OUT/refresh_extra.log 2>&1
ls $ROOT/gpurun_out/prof_r06b | wc -l
