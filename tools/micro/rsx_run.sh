#!/bin/bash
# the V1 vocoder's streaming-ResBlock launches under each cut-out: tools/micro/rsx_run.sh <mask> ...
cd "$(dirname "$0")/../.."
for m in "$@"; do
  B=tools/micro/rsx_bench_$m
  $B 32 3 3 2; $B 32 7 3 3; $B 32 11 3 1
  $B 64 3 3 2; $B 64 7 2 0; $B 64 7 1 3; $B 64 11 2 0; $B 64 11 1 1
done
