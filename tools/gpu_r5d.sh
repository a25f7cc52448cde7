#!/bin/bash
# round 5, GPU call D: DMA-staged flash attention (micro-bench + tests + FS2 line), reference-written checkpoint test
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5d; mkdir -p $OUT
cd $ROOT
( cd tools/micro; ./fa_bench_0; ./fa_bench_0p ) > $OUT/fa_bench.txt 2>&1; cat $OUT/fa_bench.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "attention or fs2 or fastspeech or reference_written or tiny_utter" > $OUT/pytest_sel.txt 2>&1; tail -15 $OUT/pytest_sel.txt
timeout 300 python bench.py --decoder fastspeech2 --no-cpu-baseline --steps 50 > $OUT/bench_fs2dec.json 2> $OUT/bench_fs2dec.err
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --decoder fastspeech2 --set front_overlap=0 > /dev/null 2>&1; cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_fs2dec_serial.csv )
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5d/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline",{})
        print(os.path.basename(f), round(j["ms_per_step"],3), r.get("kernel"), round(r.get("frac",0),4), {k:round(v,2) for k,v in (j.get("stage_ms_one_step_alone") or {}).items()})
    except Exception as e: print(f, "ERR", e)
PY
grep -i "flash\|attn" $OUT/kernel_stats_fs2dec_serial.csv | head -4
