#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "batch_flattened or decoders_alone or ragged_batch_equals or e2e_against or predicted_durations or headline or fused_f32_attention" > gpurun_out/r4_flat_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_flat_tests.log
tail -3 gpurun_out/r4_flat_tests.log
python tools/shape_log_b32.py 2> gpurun_out/r4_shape_now.txt
grep "N=1056\|N=528 \|N=1584\|N=1024" gpurun_out/r4_shape_now.txt | awk '{print $2,$3,$4,$5,$6,$9,$10,$11,$12}' | sort | uniq -c | sort -rn | head -40
for i in 1 2; do
timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_overlap=0 > gpurun_out/r4_bench_serial.json 2> gpurun_out/r4_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_serial.json").read().strip().splitlines()[-1])
print("serial", j["ms_per_step"], j["stage_ms_last_step"], [ (s["stage"], s["ms"], s["frac_mfma"]) for s in j["roofline_per_stage"] if s["stage"].startswith("decoder") or s["stage"]=="encoder"])
PY
timeout 400 python bench.py --steps 60 --no-cpu-baseline > gpurun_out/r4_bench_ovl.json 2>> gpurun_out/r4_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_ovl.json").read().strip().splitlines()[-1])
print("overlap", j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["alone"]["frac"])
PY
done
