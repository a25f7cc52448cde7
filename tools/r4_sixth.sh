#!/bin/bash
cd /root/repo
export ZVX_ERR_LOG=gpurun_out/r4_errlog.txt; rm -f $ZVX_ERR_LOG
timeout 900 python -m pytest tests -m gpu -x -q -k "batch_flattened or decoders_alone or ragged_batch_equals or e2e_against or headline or outputs_do_not_depend or converted or tiny_utterances" > gpurun_out/r4_sc_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_sc_tests.log
tail -3 gpurun_out/r4_sc_tests.log; grep "sc-fuse\|headline" $ZVX_ERR_LOG
python tools/shape_log_b32.py 2> gpurun_out/r4_shape_now.txt
grep "rows=28672" gpurun_out/r4_shape_now.txt | grep -v "N=128 \|N=256 \|N=2048\|N=1024" | awk '{print $2,$3,$4,$5,$6,$7,$9,$10,$11,$12}'
for i in 1 2; do
timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_overlap=0 > gpurun_out/r4_bench_serial.json 2> gpurun_out/r4_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_serial.json").read().strip().splitlines()[-1])
print("serial", j["ms_per_step"], j["stage_ms_last_step"], [ (s["stage"], s["ms"], s["frac_mfma"]) for s in j["roofline_per_stage"] if s["stage"].startswith("decoder")])
PY
timeout 400 python bench.py --steps 60 --no-cpu-baseline > gpurun_out/r4_bench_ovl.json 2>> gpurun_out/r4_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_ovl.json").read().strip().splitlines()[-1])
print("overlap", j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["alone"]["frac"])
PY
done
