#!/bin/bash
# round-4 first GPU pass: new tests first, A/B of the encoder planes, headline with / without the front-end overlap
cd /root/repo
export ZVX_ERR_LOG=gpurun_out/r4_errlog.txt; rm -f $ZVX_ERR_LOG
timeout 1500 python -m pytest tests -m gpu -x -q -k "predicted_durations or headline or fs2_half or pair_kernel_on_a_ragged or front_end_under or queued_calls" > gpurun_out/r4_newtests.log 2>&1; echo "newtests rc=$?" >> gpurun_out/r4_newtests.log
timeout 600 python tools/ab_encsplit.py --oracle 32 > gpurun_out/r4_ab_encsplit.txt 2>&1
timeout 400 python bench.py --steps 60 --no-cpu-baseline > gpurun_out/r4_bench_overlap.json 2> gpurun_out/r4_bench_overlap.err
timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_overlap=0 > gpurun_out/r4_bench_serial.json 2> gpurun_out/r4_bench_serial.err
timeout 400 python bench.py --steps 60 --no-cpu-baseline > gpurun_out/r4_bench_overlap2.json 2>> gpurun_out/r4_bench_overlap.err
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gputests.log
tail -3 gpurun_out/r4_newtests.log; tail -3 gpurun_out/r4_gputests.log
python - <<'PY'
import json
for f in ("overlap","serial","overlap2"):
    try:
        j=json.loads(open(f"gpurun_out/r4_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("alone",{}).get("frac"), j["stage_ms_last_step"])
    except Exception as e: print(f, "ERR", e)
PY
