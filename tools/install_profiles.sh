#!/bin/bash
# Copy the summaries of gpurun_out/prof_<tag>/ (tools/refresh_profiles.sh) into profiles/ under round-prefixed names.
#   usage: tools/install_profiles.sh <tag> <round-prefix>      e.g.  tools/install_profiles.sh r03a r03
set -eu
TAG=$1; R=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/prof_$TAG; D=$ROOT/profiles
for f in $S/bench_*.json $S/kernel_trace_*.txt $S/rocprofv3_kernel_stats_*.csv $S/pmc_*.txt $S/power_*.txt; do
  [ -s "$f" ] && cp $f $D/${R}_$(basename $f)
done
for f in $S/traffic*.json; do [ -s "$f" ] && cp $f $D/$(basename $f); done
ls -la $D
