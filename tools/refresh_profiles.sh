#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines + rocprofv3 kernel trace + PMC passes of the SAME command at the SAME sources.
# Results land in gpurun_out/prof_<tag>/ ; copy them into profiles/ (tools/install_profiles.sh <tag>) and commit.
#   usage: tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python $ROOT/bench.py --config 4 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python $ROOT/bench.py --config 4 --batch 1 --no-cpu-baseline > $OUT/bench_cfg4_b1.json 2>> $OUT/bench_cfg4.err
python $ROOT/bench.py --config 5 --steps 20 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python $ROOT/bench.py --decoder fastspeech2 --no-cpu-baseline > $OUT/bench_fs2dec.json 2> $OUT/bench_fs2dec.err
python $ROOT/bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2.json 2> $OUT/bench_v2.err
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- $CMD > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_trace_bench_n1.txt
# the tool's own --stats table of the same command
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $CMD > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$C; timeout 600 rocprofv3 --pmc $C -d /tmp/pm_$C -o pm -- $CMD > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/pm_$C -name "*.db" | head -1) --pmc > $OUT/pmc_${C}_bench_n1.txt
done
rm -rf /tmp/pm_m; timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pm_m -o pm -- $CMD > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/pm_m -name "*.db" | head -1) --pmc > $OUT/pmc_mfma_bench_n1.txt
python $ROOT/tools/make_traffic_json.py $(find /tmp/pm_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*.db" | head -1) $OUT/traffic.json 2 $(find /tmp/pm_m -name "*.db" | head -1) > $OUT/traffic_json.log 2>&1
# the bench line again, now WITH the matching traffic.json in place (bench.py quotes it only when src_sha16 agrees)
cp $OUT/traffic.json $ROOT/profiles/traffic.json
python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
ls -la $OUT
