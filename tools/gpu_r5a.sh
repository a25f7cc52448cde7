#!/bin/bash
# round 5, GPU call A: new flash attention (tests + FS2 line + trace), FP16_OVFL probe, CU-mask spatial split A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5a; mkdir -p $OUT
cd $ROOT
timeout 60 tools/micro/f16ovfl > $OUT/f16ovfl.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or fs2 or fastspeech" > $OUT/pytest_attn.txt 2>&1; tail -3 $OUT/pytest_attn.txt
timeout 300 python bench.py --decoder fastspeech2 --no-cpu-baseline --steps 50 > $OUT/bench_fs2dec.json 2> $OUT/bench_fs2dec.err
timeout 300 python bench.py --decoder fastspeech2 --no-cpu-baseline --steps 50 --set front_overlap=0 > $OUT/bench_fs2dec_serial.json 2>> $OUT/bench_fs2dec.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --decoder fastspeech2 --set front_overlap=0 > /dev/null 2>&1; cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_fs2dec_serial.csv )
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for N in 16 32 48; do for M in 0 1; do
  ZVX_CU_SPLIT=$N ZVX_CU_SPLIT_MODE=$M timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1_split${N}_m$M.json 2> $OUT/bench_n1_split${N}_m$M.err
done; done
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1_again.json 2>> $OUT/bench_n1.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline",{})
    print(sys.argv[1].split('/')[-1], round(j["ms_per_step"],3), r.get("kernel"), round(r.get("frac",0),4), (r.get("alone") or {}).get("frac"), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cat $OUT/f16ovfl.txt
grep -i "flash\|attn" $OUT/kernel_stats_fs2dec_serial.csv | head
