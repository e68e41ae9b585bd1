# part 2: rocprofv3 kernel statistics + PMC traffic of the default fp32 and bf16 steps, configs[4] ablation, final code
set -u -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/collect_profile.sh r4g > gpurun_out/r4g_collect.log 2>&1 && echo collected fp32
bash scripts/collect_profile.sh r4g_bf16 --dtype bf16 > gpurun_out/r4g_bf16_collect.log 2>&1 && echo collected bf16
bash scripts/dbg/ablation.sh r4g
