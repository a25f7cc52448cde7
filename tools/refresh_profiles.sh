#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines + rocprofv3 kernel trace + PMC passes of the SAME command at the SAME sources.
# Results land in gpurun_out/prof_<tag>/ ; copy them into profiles/ (tools/install_profiles.sh <tag>) and commit.
#   usage: tools/refresh_profiles.sh <tag> [quick|extra]     (extra = PMC passes for the secondary workloads only: other decoder / vocoders, single requests)
set -u
TAG=${1:-r04}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

# one workload: kernel trace (+ the tool's own --stats table) and the PMC passes (FETCH / WRITE / MFMA-busy in separate runs)
profile_workload() {   # <name> <bench options...>
  local NAME=$1; shift
  local OPTS="$*"
  local CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $OPTS"
  rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $CMD > /dev/null 2>&1
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_$NAME.csv
  rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- $CMD > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/kt2 -name "*.db" | head -1) > $OUT/kernel_trace_$NAME.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$C; timeout 600 rocprofv3 --pmc $C -d /tmp/pm_$C -o pm -- $CMD > /dev/null 2>&1
    python $ROOT/tools/rocpd_summary.py $(find /tmp/pm_$C -name "*.db" | head -1) --pmc > $OUT/pmc_${C}_$NAME.txt
    # per launch shape (kernel + grid): which convolutions of the step carry a variant's traffic
    python $ROOT/tools/rocpd_summary.py $(find /tmp/pm_$C -name "*.db" | head -1) --pmc-grids > $OUT/pmc_${C}_by_grid_$NAME.txt
  done
  rm -rf /tmp/pm_m; timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pm_m -o pm -- $CMD > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/pm_m -name "*.db" | head -1) --pmc > $OUT/pmc_mfma_$NAME.txt
  local TJ=traffic.json; [ "$NAME" != "bench_n1" ] && TJ=traffic_$NAME.json
  python $ROOT/tools/make_traffic_json.py $(find /tmp/pm_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*.db" | head -1) $OUT/$TJ "$OPTS" $(find /tmp/pm_m -name "*.db" | head -1) > $OUT/traffic_json_$NAME.log 2>&1
  cp $OUT/$TJ $ROOT/profiles/$TJ          # the bench line below quotes it (same sources, same workload)
}

# kernel trace only (no PMC passes): the serial schedule of the headline, whose per-launch durations are the kernels' own
trace_only() {   # <name> <bench options...>
  local NAME=$1; shift
  local CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
  rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $CMD > /dev/null 2>&1
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_$NAME.csv
  rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- $CMD > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/kt2 -name "*.db" | head -1) > $OUT/kernel_trace_$NAME.txt
}

if [ "$QUICK" = "extra" ]; then
  profile_workload fs2dec --decoder fastspeech2
  python $ROOT/bench.py --decoder fastspeech2 --no-cpu-baseline > $OUT/bench_fs2dec.json 2> $OUT/bench_fs2dec.err
  profile_workload v2 --vocoder v2
  python $ROOT/bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2.json 2> $OUT/bench_v2.err
  python $ROOT/bench.py --vocoder v2 --set front_overlap=0 --no-cpu-baseline > $OUT/bench_v2_serial.json 2> $OUT/bench_v2_serial.err
  profile_workload v3 --vocoder v3
  python $ROOT/bench.py --vocoder v3 --no-cpu-baseline > $OUT/bench_v3.json 2> $OUT/bench_v3.err
  python $ROOT/bench.py --vocoder v3 --set rb2fuse=0 --no-cpu-baseline > $OUT/bench_v3_unfused.json 2> $OUT/bench_v3_unfused.err          # A/B (round 6): every convolution of a ResBlock2 its own launch
  python $ROOT/bench.py --decoder fastspeech2 --set dec_y16=0 --no-cpu-baseline > $OUT/bench_fs2dec_y32.json 2> $OUT/bench_fs2dec_y32.err  # A/B (round 6): f32 pre-norm sums
  profile_workload b1_t64 --batch 1 --phonemes 64
  python $ROOT/bench.py --batch 1 --phonemes 64 --no-cpu-baseline > $OUT/bench_b1_t64.json 2> $OUT/bench_b1_t64.err
  profile_workload cfg4_b1 --config 4 --batch 1
  python $ROOT/bench.py --config 4 --batch 1 --no-cpu-baseline > $OUT/bench_cfg4_b1.json 2> $OUT/bench_cfg4_b1.err
  ls -la $OUT
  exit 0
fi
profile_workload bench_n1
trace_only bench_n1_serial --set front_overlap=0
python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python $ROOT/bench.py --exact-encoder --no-cpu-baseline > $OUT/bench_n1_exact_encoder.json 2> $OUT/bench_n1_exact_encoder.err
python $ROOT/bench.py --host-out --no-cpu-baseline > $OUT/bench_n1_host_out.json 2> $OUT/bench_n1_host_out.err                       # round 6: ZVX_HOST_ASYNC (pinned slots + copy stream)
python $ROOT/bench.py --host-out --host-out-sync --no-cpu-baseline > $OUT/bench_n1_host_out_sync.json 2> $OUT/bench_n1_host_out_sync.err   # ... against round 5's form: every call waits for its own copy
python $ROOT/bench.py --precision f32 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_n1_f32.json 2> $OUT/bench_n1_f32.err            # the reference's arithmetic (exact-f32 MFMA everywhere)
python $ROOT/bench.py --set front_overlap=0 --no-cpu-baseline > $OUT/bench_n1_serial.json 2> $OUT/bench_n1_serial.err                 # A/B: every call's front end behind the previous vocoder
python $ROOT/bench.py --set enc_split=1 --no-cpu-baseline > $OUT/bench_n1_bf16_planes.json 2> $OUT/bench_n1_bf16_planes.err          # A/B: the encoder's split products on bf16 planes (rounds 2-3)
python $ROOT/bench.py --in-flight 2 --no-cpu-baseline > $OUT/bench_n1_in_flight2.json 2> $OUT/bench_n1_in_flight2.err
python $ROOT/bench.py --set voc_f16=0 --no-cpu-baseline > $OUT/bench_n1_voc_bf16.json 2> $OUT/bench_n1_voc_bf16.err                    # A/B (round 5): the bf16 vocoder kernels of rounds 1-4
python $ROOT/bench.py --set voc_f16_stages=31 --no-cpu-baseline > $OUT/bench_n1_voc_half.json 2> $OUT/bench_n1_voc_half.err             # A/B (round 6): IEEE half in every stage (round 5's default)
python $ROOT/bench.py --no-cpu-baseline > $OUT/bench_n1_again.json 2> $OUT/bench_n1_again.err                                         # ... and the default once more right behind it (same box, minutes apart)
for V in 1 0; do timeout 120 python $ROOT/tools/power_readout.py --what vocoder --voc-f16 $V > $OUT/power_vocoder_f16_$V.txt 2>&1; done   # J per vocoder pass, half against bf16
timeout 120 python $ROOT/tools/power_readout.py --what step > $OUT/power_step.txt 2>&1
if [ -z "$QUICK" ]; then
  profile_workload cfg4 --config 4
  python $ROOT/bench.py --config 4 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
  python $ROOT/bench.py --config 4 --batch 1 --no-cpu-baseline > $OUT/bench_cfg4_b1.json 2>> $OUT/bench_cfg4.err
  profile_workload cfg5 --config 5
  python $ROOT/bench.py --config 5 --steps 20 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
  python $ROOT/bench.py --decoder fastspeech2 --no-cpu-baseline > $OUT/bench_fs2dec.json 2> $OUT/bench_fs2dec.err
  python $ROOT/bench.py --vocoder v2 --no-cpu-baseline > $OUT/bench_v2.json 2> $OUT/bench_v2.err
  python $ROOT/bench.py --vocoder v3 --no-cpu-baseline > $OUT/bench_v3.json 2> $OUT/bench_v3.err
  python $ROOT/bench.py --batch 1 --phonemes 64 --no-cpu-baseline > $OUT/bench_b1_t64.json 2> $OUT/bench_b1_t64.err
fi
ls -la $OUT
