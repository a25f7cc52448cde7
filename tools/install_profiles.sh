#!/bin/bash
# Copy the summaries of gpurun_out/prof_<tag>/ (tools/refresh_profiles.sh) into profiles/ under round-prefixed names.
#   usage: tools/install_profiles.sh <tag> <round-prefix>      e.g.  tools/install_profiles.sh r02a r02
set -eu
TAG=$1; R=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/prof_$TAG; D=$ROOT/profiles
cp $S/bench_n1.json $D/${R}_bench_n1.json
cp $S/bench_cfg4.json $D/${R}_bench_cfg4.json
cp $S/bench_cfg4_b1.json $D/${R}_bench_cfg4_b1.json
cp $S/bench_cfg5.json $D/${R}_bench_cfg5.json
cp $S/bench_fs2dec.json $D/${R}_bench_fs2_decoder.json
cp $S/bench_v2.json $D/${R}_bench_hifigan_v2.json
cp $S/kernel_trace_bench_n1.txt $D/${R}_kernel_trace_bench_n1.txt
cp $S/rocprofv3_kernel_stats.csv $D/${R}_rocprofv3_kernel_stats_bench_n1.csv
cp $S/pmc_FETCH_SIZE_bench_n1.txt $D/${R}_pmc_fetch_bench_n1.txt
cp $S/pmc_WRITE_SIZE_bench_n1.txt $D/${R}_pmc_write_bench_n1.txt
cp $S/pmc_mfma_bench_n1.txt $D/${R}_pmc_mfma_bench_n1.txt
cp $S/traffic.json $D/traffic.json
ls -la $D
