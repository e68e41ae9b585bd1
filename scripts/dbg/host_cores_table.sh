#!/bin/bash
# Run ON THE GPU BOX: how many host cores does one rank need before the step turns host-bound?  
#   bash scripts/dbg/host_cores_table.sh <tag>   -> gpurun_out/<tag>_host_cores.txt (one JSON-derived line per K and dtype)
set -u -o pipefail
: "${GRAFT_REPO_ROOT:?runs on the GPU box}"
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r4}
OUT=gpurun_out/${TAG}_host_cores.txt
: > "$OUT"
rc=0
for dt in f32 bf16; do
  for k in 16 8 4 2 1; do
    line=$(python bench.py --dtype $dt --host-cores $k --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    if [ -z "$line" ]; then echo "$dt K=$k FAILED" >> "$OUT"; rc=1; continue; fi
    echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%s K=%2d cores=%2d ms_per_step=%.3f median=%.3f host_enqueue_ms=%.3f frames_per_s=%.0f' % ('$dt', $k, d['host_cores'], d['ms_per_step'], d['ms_per_step_median'], d['host_enqueue_ms_per_step'], d['value']))" >> "$OUT"
  done
done
cat "$OUT"
exit $rc
