#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's closing validation in one call.
#   TAG=r3k [SKIP_TESTS=1] bash scripts/dbg/validate_round.sh
# full GPU suite -> gpurun_out/<TAG>_tests.txt; rocprofv3 sets of the fp32 and bf16 step (scripts/collect_profile.sh <TAG>,
# <TAG>_bf16), the instruction-mix PMC pass and the streaming kernel stats -> gpurun_out/prof_<TAG>*/; the default bench lines
# -> gpurun_out/<TAG>_bench.txt, <TAG>_bench_bf16.txt.  Summaries for profiles/: scripts/summarize_profile.py <TAG> <name> "<title>".
set -u -o pipefail
TAG=${TAG:-rX}
SKIP_TESTS=${SKIP_TESTS:-}
: "${GRAFT_REPO_ROOT:?validate_round.sh runs on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
rc=0
step() {   # step <label> <command...>: run, report, remember a failure (the script's exit status is the OR of all steps)
    local label=$1; shift
    if "$@"; then echo "$label ok"; else echo "$label FAILED ($?)"; rc=1; fi
}
if [ -z "$SKIP_TESTS" ]; then
    timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "gpurun_out/${TAG}_tests.txt" 2>&1
    trc=$?; echo "rc=$trc" >> "gpurun_out/${TAG}_tests.txt"; tail -3 "gpurun_out/${TAG}_tests.txt"
    [ "$trc" = 0 ] || exit 1           # a red or killed GPU step: start no further GPU step in this call
fi
collect() { bash scripts/collect_profile.sh "$@" > "gpurun_out/$1_collect.log" 2>&1; }
step "profile fp32" collect "${TAG}"
step "profile bf16" collect "${TAG}_bf16" --dtype bf16
cd /tmp && export TMPDIR=/tmp
P="$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}"
mkdir -p "$P"
B="$GRAFT_REPO_ROOT/bench.py"
insts() { rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d "$P/insts" -o i -- python3 "$B" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$P/insts.log" 2>&1; }
stream() { rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stream$1" -o s -- python3 "$B" --streaming $2 --steps 300 --no-cpu-baseline > "$P/stream$1.log" 2>&1; }
step "instruction mix" insts
step "stream stats" stream "" ""
step "stream tgru stats" stream "_tgru" "--tgru"
find "$GRAFT_REPO_ROOT/gpurun_out" -name "*kernel_trace.csv" -size +20M -delete
cd "$GRAFT_REPO_ROOT" || exit 1
bench() { python bench.py "${@:2}" > "gpurun_out/${TAG}_$1.txt" 2> "gpurun_out/${TAG}_$1.err"; }
step "bench fp32" bench bench; tail -c 600 "gpurun_out/${TAG}_bench.txt"
# per-launch times and flops of the fp32 step (profiles/*_roofline.md): its own run, the extras of the default line would overwrite the log
launchlog() { TRUNET_BENCH_LAUNCH_LOG="$GRAFT_REPO_ROOT/gpurun_out/${TAG}_launch_f32.log" python bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1; }
step "launch log" launchlog
step "bench bf16" bench bench_bf16 --dtype bf16 --no-cpu-baseline; tail -c 400 "gpurun_out/${TAG}_bench_bf16.txt"
exit $rc
