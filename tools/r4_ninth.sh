#!/bin/bash
cd /root/repo
for k in 11 7 3; do tools/micro/ps_bench_0_old $k 5; tools/micro/ps_bench_0 $k 5; tools/micro/ps_bench_0_old $k 5 32 57344 40 1; tools/micro/ps_bench_0 $k 5 32 57344 40 1; done
tools/micro/ps_bench_0p 3 5; tools/micro/ps_bench_0p 11 5
timeout 900 python -m pytest tests -m gpu -x -q -k "pair_kernel or reproducible or config4 or vocoder_v1 or batch_invariance" 2>&1 | grep -v "^Host\|^Libr\|^ROCm" | tail -3
timeout 600 python tools/stress_pairstream.py 40 2>&1 | tail -3
timeout 600 python tools/race_hunt.py 150 2>&1 | tail -3
