#!/bin/bash
# ablation of pw_bwd_kernel<..., X3> on the GPU box: scripts/dbg/ab/lib_pwb<PWB_ABL>.so (built in the container)
cd "$(dirname "$0")/../.."
for w in enc dec; do
  echo "== $w"
  TRUNET_GEMM_X3=0 TRUNET_HIP_LIB=scripts/dbg/ab/lib_pwb0.so timeout -k 10 100 python scripts/ubench_pwbwd.py $w 5 2>/dev/null | sed 's/^/fp32-MFMA  /'
  for abl in 0 4 8 16 32 48; do
    TRUNET_HIP_LIB=scripts/dbg/ab/lib_pwb$abl.so timeout -k 10 100 python scripts/ubench_pwbwd.py $w 5 2>/dev/null | sed "s/^/x3 abl=$abl  /"
  done
done
