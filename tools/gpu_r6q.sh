#!/bin/bash
# round 6, GPU call Q: the race screens at the head sources
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6q; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 900 python tools/race_hunt.py 120 > $OUT/race_hunt.txt 2>&1; cat $OUT/race_hunt.txt | cut -c1-160
timeout 900 python tools/race_hunt_all.py > $OUT/race_hunt_all.txt 2>&1; tail -3 $OUT/race_hunt_all.txt | cut -c1-160
python -c "import bench; print('src_sha16', bench.src_sha16())" >> $OUT/race_hunt.txt
