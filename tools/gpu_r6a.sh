#!/bin/bash
# round 6, GPU call A: baseline at the round's first sources -- per-launch shape logs of the front end (headline, V2, V3, FS2) + the default line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6a; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
timeout 300 python tools/shape_log_b32.py > $OUT/shape_v1.txt 2>&1
ZVX_VOCODER=v3 timeout 300 python tools/shape_log_b32.py > $OUT/shape_v3.txt 2>&1
ZVX_VOCODER=v2 timeout 300 python tools/shape_log_b32.py > $OUT/shape_v2.txt 2>&1
ZVX_DECODER=fastspeech2 timeout 300 python tools/shape_log_b32.py > $OUT/shape_fs2.txt 2>&1
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 python bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2.json 2>> $OUT/bench_n1.err
timeout 300 python bench.py --host-out --no-cpu-baseline > $OUT/bench_host.json 2>> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.json | head -c 300; echo
grep -c launch $OUT/shape_v1.txt
