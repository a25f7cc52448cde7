#!/bin/bash
# builds tools/micro/ps_bench_<mask>[p] for the given PS_EXP masks (suffix p: with -DPS_PROFILE)
cd "$(dirname "$0")/../.."
for m in "$@"; do
  prof=""; mask=$m
  case $m in *p) prof="-DPS_PROFILE"; mask=${m%p};; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -Wno-inline-asm -Wno-unused-value -I zerovox_amd/csrc -DPS_EXP=$mask $prof $PS_DEFS tools/micro/ps_bench.hip -o tools/micro/ps_bench_$m$PS_TAG &
done
wait
ls -la tools/micro/ps_bench_*
