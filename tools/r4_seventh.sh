#!/bin/bash
cd /root/repo
export ZVX_ERR_LOG=gpurun_out/r4_errlog.txt; rm -f $ZVX_ERR_LOG
timeout 1200 python -m pytest tests -m gpu -x -q -k "batch_flattened or decoders_alone or ragged_batch_equals or e2e_against or headline or fs2 or fused_attention or long_sequences or one_shot or secondary" > gpurun_out/r4_fs2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_fs2_tests.log
tail -3 gpurun_out/r4_fs2_tests.log
for fl in 0 1; do
timeout 400 python bench.py --steps 60 --no-cpu-baseline --decoder fastspeech2 --set front_overlap=0 --set dec_flat=$fl > gpurun_out/r4_bench_fs2_$fl.json 2> gpurun_out/r4_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_fs2_$fl.json").read().strip().splitlines()[-1])
print("fs2 serial flat=$fl", j["ms_per_step"], j["stage_ms_last_step"], [ (s["stage"], s["ms"], s["frac_mfma"]) for s in j["roofline_per_stage"] if s["stage"].startswith("decoder")])
PY
done
timeout 400 python bench.py --steps 60 --no-cpu-baseline --decoder fastspeech2 > gpurun_out/r4_bench_fs2.json 2> gpurun_out/r4_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r4_bench_fs2.json").read().strip().splitlines()[-1])
print("fs2 overlap", j["ms_per_step"])
PY
