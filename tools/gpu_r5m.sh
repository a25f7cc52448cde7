#!/bin/bash
# round 5, GPU call M: speaker encoder tests + config 5 repeats + kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5m; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
ZVX_ERR_LOG=$OUT/errlog.txt timeout 1200 python -m pytest tests -m gpu -x -q -k "speaker or spk or embed or resnet or refckpt or reference_written or determin" > $OUT/pytest_sel.txt 2>&1; tail -4 $OUT/pytest_sel.txt
grep -h variants $OUT/errlog.txt
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --config 5 > $OUT/bench_cfg5_new_$i.json 2>> $OUT/err.txt
  timeout 300 python bench.py --no-cpu-baseline --config 5 --set slab_small=34 --set spk_pool_fuse=0 --set spk_s2_fuse=0 > $OUT/bench_cfg5_old_$i.json 2>> $OUT/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5m/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(j["ms_per_step"],3), round(j["value"]), [(k["name"],k["launches"],k["ms"]) for k in j.get("kernels_one_step",[]) if "conv2d" in k["name"] or "convreg" in k["name"]])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --config 5 > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/kt2 -name "*.db" | head -1) > $OUT/kernel_trace_new.txt
head -22 $OUT/kernel_trace_new.txt | cut -c1-150
