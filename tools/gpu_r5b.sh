#!/bin/bash
# round 5, GPU call B: pipelined flash attention
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5b; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or fs2 or fastspeech" > $OUT/pytest_attn.txt 2>&1; tail -3 $OUT/pytest_attn.txt
timeout 300 python bench.py --decoder fastspeech2 --no-cpu-baseline --steps 50 > $OUT/bench_fs2dec.json 2> $OUT/bench_fs2dec.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --decoder fastspeech2 --set front_overlap=0 > /dev/null 2>&1; cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_fs2dec_serial.csv )
for C in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm_$C && timeout 600 rocprofv3 --pmc $C -d /tmp/pm_$C -o pm -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --decoder fastspeech2 --set front_overlap=0 > /dev/null 2>&1; python $ROOT/tools/rocpd_summary.py $(find /tmp/pm_$C -name "*.db" | head -1) --pmc > $OUT/pmc_${C}_fs2dec.txt ); done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5b/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline",{})
    print(os.path.basename(f), round(j["ms_per_step"],3), r.get("kernel"), round(r.get("frac",0),4), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items()})
PY
grep -i "flash\|attn" $OUT/kernel_stats_fs2dec_serial.csv | head
grep -i "flash" $OUT/pmc_*_fs2dec.txt | head
