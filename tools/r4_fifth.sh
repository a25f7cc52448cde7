#!/bin/bash
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gputests.log
tail -3 gpurun_out/r4_gputests.log
timeout 300 python bench.py --config 5 --steps 40 --no-cpu-baseline > gpurun_out/r4_cfg5.json 2> gpurun_out/r4_cfg5.err
timeout 300 python bench.py --vocoder v2 --steps 60 --no-cpu-baseline > gpurun_out/r4_v2.json 2> gpurun_out/r4_v2.err
timeout 300 python bench.py --vocoder v3 --steps 60 --no-cpu-baseline > gpurun_out/r4_v3.json 2> gpurun_out/r4_v3.err
timeout 300 python bench.py --decoder fastspeech2 --steps 60 --no-cpu-baseline > gpurun_out/r4_fs2.json 2> gpurun_out/r4_fs2.err
timeout 300 python bench.py --config 4 --steps 60 --no-cpu-baseline > gpurun_out/r4_cfg4.json 2> gpurun_out/r4_cfg4.err
timeout 300 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r4_head.json 2> gpurun_out/r4_head.err
python - <<'PY'
import json
for f in ("cfg5","v2","v3","fs2","cfg4","head"):
    try:
        j=json.loads(open(f"gpurun_out/r4_{f}.json").read().strip().splitlines()[-1])
        print(f, round(j["ms_per_step"],3), round(j["value"],1), j["roofline"]["kernel"], round(j["roofline"]["frac"],3))
        if f in ("cfg5",):
            for k in j["kernels_one_step"]: print("   ", k)
            for k in j["roofline_per_stage"]: print("   ", k)
    except Exception as e: print(f, "ERR", e)
PY
