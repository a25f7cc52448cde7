#!/bin/bash
# round 5, GPU call K: kernel traces of config 5 (speaker encoder): default / SE pool as its own pass / previous convolution kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5k; mkdir -p $OUT; rm -f $OUT/*
cd /tmp && export TMPDIR=/tmp
tr() { # name opts
  local NAME=$1; shift
  rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --config 5 $* > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/kt2 -name "*.db" | head -1) > $OUT/kernel_trace_$NAME.txt
  head -16 $OUT/kernel_trace_$NAME.txt | cut -c1-150
}
tr new
tr nopool --set spk_pool_fuse=0
tr old --set slab_small=34
