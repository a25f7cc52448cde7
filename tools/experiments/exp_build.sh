#!/bin/bash
# Dev aid: build libzvx variants with -DZVX_EXP=<mask> in gemm.hip (timing experiments; results are WRONG by design).
#   tools/exp_build.sh 1 2 4 ...   -> zerovox_amd/libzvx_exp<mask>.so
set -e
cd "$(dirname "$0")/.."
python -m zerovox_amd.build >/dev/null
C=zerovox_amd/csrc
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm -fno-honor-nans -DZVX_EXP=$m -c $C/gemm.hip -o $C/gemm_exp$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/gemm_exp$m.o $C/resstream.o $C/attention.o $C/ops.o $C/zvx.o -ldl -o zerovox_amd/libzvx_exp$m.so && echo built exp$m ) &
done
wait
