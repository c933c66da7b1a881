mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
(timeout 600 python -m pytest tests/test_reference_surface_gpu.py "tests/test_properties_gpu.py::test_fast_stream_query_follows_torchs_current_stream" "tests/test_properties_gpu.py::test_composite_pair_without_the_scan_kernel_equals_the_pair_with_it" -q -p no:cacheprovider 2>&1 | tail -8) > $O/pytest.txt 2>&1
timeout 300 python tools/render_sweep_trained.py > $O/render_sweep.txt 2> $O/render_sweep.err
timeout 200 python tools/api_host_breakdown.py > $O/api_host.json 2> $O/api_host.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run > $O/bench.json 2> $O/bench.err
cat $O/render_sweep.txt; tail -3 $O/pytest.txt; python -c "
import json; d=json.load(open('$O/api_host.json')); print({k:(v['ms_per_step'],v['host_us']) for k,v in d.items()}); b=json.load(open('$O/bench.json')); print(b['ms_per_step'], b['api_path']['ms_per_step'], b['api_path_plain']['ms_per_step'], [ (s['stage'],s['ms']) for s in b['roofline']['stages']])"
