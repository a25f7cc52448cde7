#!/bin/bash
# The headline step with every host core kept busy by other processes (2 spinning processes per core): a queued zvx_synthesize
# (forced durations, device output, no_sync) does not wait on the host, so the line should barely move.
#   usage (GPU box): tools/host_contention.sh [bench options...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$(( $(nproc) * 2 ))
echo "quiet host:"; python $ROOT/bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  ms_per_step', round(d['ms_per_step'],3))"
PIDS=""
for i in $(seq $N); do python -c "
import time
t=time.time()
while time.time()-t<90: pass" & PIDS="$PIDS $!"; done
sleep 1
echo "$N spinning processes on $(nproc) cores:"; python $ROOT/bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  ms_per_step', round(d['ms_per_step'],3))"
for p in $PIDS; do kill $p 2>/dev/null; done
wait 2>/dev/null
