run() { # label, env
  fails=0
  for i in 1 2 3 4 5 6; do
    out=$(env $2 timeout -k 10 120 python -m pytest "tests/test_network_gpu.py::test_forward_backward_vs_oracle_f64[300]" -x -q 2>&1 | grep -E "^E  |passed|failed" | head -3)
    case "$out" in *failed*) fails=$((fails+1)); echo "$1 run $i: $out" | head -3;; esac
  done
  echo "$1: $fails / 6 failed"
}
run current "X=1"
run nofused "TRUNET_FUSED_PWBWD=0"
run nowide "TRUNET_GEMM_WIDE=0"
