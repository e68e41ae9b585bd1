# round-4 closing validation after the split encoder of the streaming kernel: whole GPU suite, smoke, default bench line, kernel statistics
# of the streaming launches (stateless and stateful)
set -u -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4e_tests.txt 2>&1; rc=$?; echo rc=$rc >> gpurun_out/r4e_tests.txt; tail -3 gpurun_out/r4e_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r4e_bench_full.txt 2>gpurun_out/r4e_bench_full.err; echo bench rc=$?
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4e_stream -o s -- python3 $R/bench.py --streaming --steps 300 --no-cpu-baseline > $R/gpurun_out/r4e_stream_prof.log 2>&1; echo prof rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4e_stream_tgru -o s -- python3 $R/bench.py --streaming --tgru --steps 300 --no-cpu-baseline > $R/gpurun_out/r4e_stream_tgru_prof.log 2>&1; echo prof rc=$?
cd $R
find gpurun_out/prof_r4e_stream gpurun_out/prof_r4e_stream_tgru -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4e_bench_full.txt").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"]["mfma"], d["roofline"]["kernel"], d["roofline"]["bound"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k: (v.get("ms_per_step") or v.get("value") or v) for k, v in d.get("other_configs", {}).items()})
PY
