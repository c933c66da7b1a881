mkdir -p gpurun_out/r04h
O=gpurun_out/r04h
timeout 300 python tools/api_cprofile.py > $O/api_cprofile.txt 2> $O/api_cprofile.err
(timeout 600 python -m pytest tests/test_bench_gpu.py -q -p no:cacheprovider -k "one_rank" 2>&1 | tail -8) > $O/pytest.txt 2>&1
head -75 $O/api_cprofile.txt | cut -c1-160; tail -4 $O/pytest.txt
