#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q -k "full_size or headline or config2 or config3 or batch_flattened or rccl_gather_path_on_a or front_end_under" > gpurun_out/r4_split_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_split_tests.log
tail -3 gpurun_out/r4_split_tests.log
python tools/shape_log_b32.py 2> gpurun_out/r4_shape_now.txt
grep "rows=28672" gpurun_out/r4_shape_now.txt | grep "N=528 \|N=512 \|N=16 " | awk '{print $2,$3,$4,$5,$6,$7,$9,$10,$11,$12}'
for ss in 2 34 2 34; do
timeout 400 python bench.py --steps 60 --no-cpu-baseline --set front_overlap=0 --set slab_small=$ss 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial slab_small=$ss', round(j['ms_per_step'],3), [(s['stage'], s['ms'], s['frac_mfma']) for s in j['roofline_per_stage'] if s['stage']=='decoder'])"
done
for ss in 2 34; do
timeout 400 python bench.py --steps 60 --no-cpu-baseline --set slab_small=$ss 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap slab_small=$ss', round(j['ms_per_step'],3))"
done
