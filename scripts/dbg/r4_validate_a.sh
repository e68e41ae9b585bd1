timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4_t7.txt 2>&1; echo rc=$? >> gpurun_out/r4_t7.txt; tail -6 gpurun_out/r4_t7.txt
bash scripts/dbg/ablation.sh r4
python bench.py --streaming --no-cpu-baseline > gpurun_out/r4_stream.txt 2>/dev/null; tail -c 300 gpurun_out/r4_stream.txt
python bench.py --streaming --audio > gpurun_out/r4_stream_audio.txt 2>gpurun_out/r4_stream_audio.err; cat gpurun_out/r4_stream_audio.txt | cut -c1-300
python bench.py --streaming --audio --tgru > gpurun_out/r4_stream_audio_tgru.txt 2>/dev/null; cat gpurun_out/r4_stream_audio_tgru.txt | cut -c1-200
