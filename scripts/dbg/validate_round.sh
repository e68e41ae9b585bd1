#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's closing validation in one call.
#   TAG=r3k [SKIP_TESTS=1] bash scripts/dbg/validate_round.sh
# full GPU suite -> gpurun_out/<TAG>_tests.txt; rocprofv3 sets of the fp32 and bf16 step (scripts/collect_profile.sh <TAG>,
# <TAG>_bf16), the instruction-mix PMC pass and the streaming kernel stats -> gpurun_out/prof_<TAG>*/; the default bench lines
# -> gpurun_out/<TAG>_bench.txt, <TAG>_bench_bf16.txt.  Summaries for profiles/: scripts/summarize_profile.py <TAG> <name> "<title>".
set -o pipefail
TAG=${TAG:-rX}
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1; echo rc=$? >> gpurun_out/${TAG}_tests.txt; tail -3 gpurun_out/${TAG}_tests.txt
grep -q "rc=0" gpurun_out/${TAG}_tests.txt || exit 1
fi
bash scripts/collect_profile.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1 && echo collected fp32
bash scripts/collect_profile.sh ${TAG}_bf16 --dtype bf16 > gpurun_out/${TAG}_bf16_collect.log 2>&1 && echo collected bf16
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $P/insts -o i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $P/insts.log 2>&1 && echo insts done
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stream -o s -- python3 $GRAFT_REPO_ROOT/bench.py --streaming --steps 300 --no-cpu-baseline > $P/stream.log 2>&1 && echo stream done
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stream_tgru -o s -- python3 $GRAFT_REPO_ROOT/bench.py --streaming --tgru --steps 300 --no-cpu-baseline > $P/stream_tgru.log 2>&1 && echo stream tgru done
find $GRAFT_REPO_ROOT/gpurun_out -name "*kernel_trace.csv" -size +20M -delete
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.txt 2>gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.txt
# per-launch times and flops of the fp32 step (profiles/*_roofline.md): its own run, the extras of the default line would overwrite the log
TRUNET_BENCH_LAUNCH_LOG=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_launch_f32.log python bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.txt 2>/dev/null; tail -c 400 gpurun_out/${TAG}_bench_bf16.txt
