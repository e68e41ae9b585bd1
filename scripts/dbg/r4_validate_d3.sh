# part 3: CPU baseline at the BASELINE.md protocol (50 steps), streaming lines, the forced-distributed line, bf16 line
set -u -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python bench.py --cpu-steps 50 --cpu-budget 400 --no-extras > gpurun_out/r4d_bench_cpu50.txt 2>/dev/null; echo cpu50 rc=$?
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r4d_bench_bf16.txt 2>/dev/null; echo bf16 rc=$?
python bench.py --streaming --no-cpu-baseline > gpurun_out/r4d_stream.txt 2>/dev/null
python bench.py --streaming --audio --no-cpu-baseline > gpurun_out/r4d_stream_audio.txt 2>/dev/null
python bench.py --streaming --audio --tgru --no-cpu-baseline > gpurun_out/r4d_stream_audio_tgru.txt 2>/dev/null
python bench.py --force-dist --no-cpu-baseline --no-extras > gpurun_out/r4d_force_dist.txt 2>/dev/null
python - <<'PY'
import json
for f in ("bench_cpu50", "bench_bf16", "stream", "stream_audio", "stream_audio_tgru", "force_dist"):
    try:
        d = json.loads(open("gpurun_out/r4d_%s.txt" % f).read().strip().splitlines()[-1])
        print(f, d.get("ms_per_step"), d.get("value"), d.get("unit"), d.get("cpu_baseline"), (d.get("dist") or {}).get("allreduce_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
