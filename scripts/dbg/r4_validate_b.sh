# round-4 closing validation, part b (on the GPU box): full suite, profiles of the default fp32 / bf16 steps and of the opt-in bf16x3 kind
set -u -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4_t8.txt 2>&1; echo rc=$? >> gpurun_out/r4_t8.txt; tail -6 gpurun_out/r4_t8.txt
bash scripts/collect_profile.sh r4c > gpurun_out/r4c_collect.log 2>&1 && echo collected fp32
bash scripts/collect_profile.sh r4c_bf16 --dtype bf16 > gpurun_out/r4c_bf16_collect.log 2>&1 && echo collected bf16
bash scripts/collect_profile.sh r4c_x3 --mfma bf16x3 > gpurun_out/r4c_x3_collect.log 2>&1 && echo collected x3
TRUNET_BENCH_LAUNCH_LOG=$PWD/gpurun_out/r4c_launch_f32.log python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4c_bench_noextras.txt 2>/dev/null
TRUNET_BENCH_LAUNCH_LOG=$PWD/gpurun_out/r4c_launch_x3.log python bench.py --mfma bf16x3 --no-cpu-baseline --no-extras > gpurun_out/r4c_bench_x3.txt 2>/dev/null
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r4c_bench_bf16.txt 2>/dev/null
python bench.py --force-dist --no-cpu-baseline --no-extras > gpurun_out/r4c_force_dist.txt 2>/dev/null
tail -c 300 gpurun_out/r4c_bench_x3.txt | head -c 200
