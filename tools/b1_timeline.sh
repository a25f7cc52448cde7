#!/bin/bash
# Per-dispatch timeline (rocprofv3 kernel trace) of the LAST single 64-phoneme request of a short loop: which launches are on the
# critical path, how long each runs, and the gaps between them.   usage (GPU box): tools/b1_timeline.sh <out.txt> [dispatches]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/b1_timeline.txt}; N=${2:-190}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/b1kt; timeout 600 rocprofv3 --kernel-trace -d /tmp/b1kt -o kt -- python $ROOT/tools/latency_log.py 64 > /tmp/b1kt.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/b1kt -name "*.db" | head -1) --timeline $N > $OUT 2>&1
tail -3 $OUT
