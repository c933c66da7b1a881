mkdir -p gpurun_out/r04a
(time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/r04a/pytest.txt 2>&1
(time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err)  2> gpurun_out/r04a/bench.time
(time timeout 500 bash tools/render_trained_trace.sh r04a > gpurun_out/r04a/render_trace.log 2>&1) 2> gpurun_out/r04a/render_trace.time
tail -5 gpurun_out/r04a/pytest.txt; head -c 600 gpurun_out/r04a/bench.json
