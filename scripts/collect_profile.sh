#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace statistics and the two HBM-traffic PMC passes of the default
# bench step, each in its own rocprofv3 run (counters are never combined with other trace domains).
#   bash scripts/collect_profile.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
set -e
TAG=${1:-x}
shift || true
EXTRA="$@"          # extra bench.py arguments, e.g. --dtype bf16
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > "$OUT/fetch.log" 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > "$OUT/write.log" 2>&1
echo "WRITE_SIZE pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq" -o q -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > "$OUT/sq.log" 2>&1
echo "SQ pass done"
# keep what travels back small: the per-dispatch traces are large, the summaries are what profiles/ keeps
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete || true
ls -la "$OUT"/*/* | head -30
