set -o pipefail
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t23.txt 2>&1; echo rc=$? >> gpurun_out/r3_t23.txt; tail -3 gpurun_out/r3_t23.txt
grep -q "rc=0" gpurun_out/r3_t23.txt || exit 1
fi
bash scripts/collect_profile.sh r3i > gpurun_out/r3i_collect.log 2>&1 && echo collected fp32
bash scripts/collect_profile.sh r3i_bf16 --dtype bf16 > gpurun_out/r3i_bf16_collect.log 2>&1 && echo collected bf16
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3i/insts -o i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_r3i/insts.log 2>&1 && echo insts done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3i/stream -o s -- python3 $GRAFT_REPO_ROOT/bench.py --streaming --steps 300 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r3i/stream.log 2>&1 && echo stream done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3i/stream_tgru -o s -- python3 $GRAFT_REPO_ROOT/bench.py --streaming --tgru --steps 300 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r3i/stream_tgru.log 2>&1 && echo stream tgru done
find $GRAFT_REPO_ROOT/gpurun_out -name "*kernel_trace.csv" -size +20M -delete
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r3_bench10.txt 2>gpurun_out/r3_bench10.err; tail -c 600 gpurun_out/r3_bench10.txt
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r3_bench10_bf16.txt 2>/dev/null; tail -c 400 gpurun_out/r3_bench10_bf16.txt
