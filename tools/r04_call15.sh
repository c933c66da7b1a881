(timeout 900 python -m pytest tests/test_properties_gpu.py "tests/test_ddp_gpu.py::test_sharded_exchange_trains_like_the_allreduce_exchange_on_one_gpu" -q -p no:cacheprovider 2>&1 | tail -12)
