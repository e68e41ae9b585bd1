# round-4 closing validation, final code, part 1: whole GPU suite, smoke, the default bench line
set -u -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4g_tests.txt 2>&1; rc=$?; echo rc=$rc >> gpurun_out/r4g_tests.txt; tail -3 gpurun_out/r4g_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r4g_bench_full.txt 2>gpurun_out/r4g_bench_full.err; echo bench rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4g_bench_full.txt").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"]["mfma"], d["roofline"]["kernel"], d["roofline"]["bound"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k: v for k, v in list(d["roofline"]["kernel_ms_per_step"].items())[:8]})
print({k: (v.get("ms_per_step") or v.get("value") or v) for k, v in d.get("other_configs", {}).items()})
PY
