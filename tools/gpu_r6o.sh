#!/bin/bash
# round 6, GPU call O: FS2 / SCLN decoder with one Q | K | V projection: decoder tests + FS2 line A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6o; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1200 python -m pytest tests -m gpu -q -x -k "fs2 or fastspeech2 or decoder or golden or attention or ragged or flattened or padding or long_seq" > $OUT/pytest_sel.txt 2>&1; tail -3 $OUT/pytest_sel.txt
for i in 1 2; do
timeout 300 python bench.py --decoder fastspeech2 --no-cpu-baseline > $OUT/bench_fs2_qkv1_$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --decoder fastspeech2 --set dec_qkv=0 --no-cpu-baseline > $OUT/bench_fs2_qkv0_$i.json 2>> $OUT/bench.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6o/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["ms_per_step"],3), j.get("output_ok"), {k:round(v,3) for k,v in j["stage_ms_one_step_alone"].items() if v})
    except Exception as e: print(f, "ERR", e)
PY
grep -i "fastspeech2\]\|fs2" $OUT/errlog.txt | grep mel | head -6
