#!/bin/bash
cd /root/repo
export ZVX_ERR_LOG=gpurun_out/r4_errlog.txt; rm -f $ZVX_ERR_LOG
for pr in 0 1 -1; do
  timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_prio=$pr > gpurun_out/r4_bench_prio$pr.json 2> gpurun_out/r4_bench_prio$pr.err
done
timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_overlap=0 > gpurun_out/r4_bench_serial.json 2> gpurun_out/r4_bench_serial.err
python - <<'PY'
import json
for f in ("prio0","prio1","prio-1","serial"):
    try:
        j=json.loads(open(f"gpurun_out/r4_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("alone",{}).get("frac"), j["stage_ms_last_step"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python tools/ab_encsplit.py --oracle 32 > gpurun_out/r4_ab_encsplit.txt 2>&1; tail -4 gpurun_out/r4_ab_encsplit.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gputests.log
tail -3 gpurun_out/r4_gputests.log
