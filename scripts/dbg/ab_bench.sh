#!/bin/bash
# A/B of library builds on ONE box (box-to-box variation is ~1 %): runs bench.py twice per scripts/dbg/ab/lib_*.so;
# per-launch times of the instrumented step go to gpurun_out/ab_<lib>.log
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
for lib in scripts/dbg/ab/lib_*.so; do
  nm=$(basename $lib .so)
  TRUNET_BENCH_LAUNCH_LOG=gpurun_out/ab_$nm.log TRUNET_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %8.3f ms median %8.3f' % ('$nm', d['ms_per_step'], d['ms_per_step_median']))"
done; done
