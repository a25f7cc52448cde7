#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "batch_flattened or decoders_alone or ragged_batch_equals or e2e_against or config2 or outputs_do_not_depend or full_size" > gpurun_out/r4_flat_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_flat_tests.log
tail -3 gpurun_out/r4_flat_tests.log
for fl in 0 1 0 1; do
  timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_overlap=0 --set dec_flat=$fl > gpurun_out/r4_bench_flat${fl}_serial.json 2> gpurun_out/r4_bench_flat.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_flat${fl}_serial.json").read().strip().splitlines()[-1])
print("serial dec_flat=$fl", j["ms_per_step"], j["stage_ms_last_step"]["decoder"], [ (s["stage"], s["ms"], s["frac_mfma"]) for s in j["roofline_per_stage"] if s["stage"].startswith("decoder")])
PY
done
for fl in 0 1; do
  timeout 400 python bench.py --steps 60 --no-cpu-baseline --set dec_flat=$fl > gpurun_out/r4_bench_flat${fl}.json 2>> gpurun_out/r4_bench_flat.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_flat${fl}.json").read().strip().splitlines()[-1])
print("overlap dec_flat=$fl", j["ms_per_step"])
PY
done
