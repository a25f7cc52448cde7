#!/bin/bash
# round 6, GPU call D: per-stage arithmetic of the generator: time + error per mask (two passes, alternating)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6d; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 900 python tools/ab_voc_stages.py 31 27 25 29 19 17 0 31 27 25 0 > $OUT/ab_voc_stages.txt 2>&1
cat $OUT/ab_voc_stages.txt | tail -14
