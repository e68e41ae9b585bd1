#!/bin/bash
# Run ON THE GPU BOX: BASELINE.json configs[4] -- multi-resolution STFT loss on/off x PCEN feature on/off, fp32 and bf16, on the
# final code: one bench JSON line each -> gpurun_out/<tag>_ablation.jsonl (scripts/ablation_table.py turns it, together with the
# per-kernel PMC traffic of the round's profile, into profiles/<name>_ablation.md).
set -u -o pipefail
: "${GRAFT_REPO_ROOT:?runs on the GPU box}"
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r4}
OUT=gpurun_out/${TAG}_ablation.jsonl
: > "$OUT"
rc=0
for dt in f32 bf16; do
  for fl in "" "--no-stft-loss" "--no-pcen" "--no-stft-loss --no-pcen"; do
    line=$(python bench.py --dtype $dt $fl --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    if [ -z "$line" ]; then echo "{\"error\": \"$dt $fl\"}" >> "$OUT"; rc=1; else echo "$line" >> "$OUT"; fi
    echo "$dt [$fl] done"
  done
done
exit $rc
