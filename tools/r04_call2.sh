mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
(time timeout 900 python -m pytest tests/test_ddp_gpu.py tests/test_bench_gpu.py tests/test_properties_gpu.py "tests/test_train_gpu.py::test_two_round_forward_ignores_stale_values_behind_a_stop" tests/test_reference_surface_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -40) > $O/pytest.txt 2>&1
# 1-rank RCCL group: native exchange (chunks / groups) vs torch.distributed vs no group
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-api --no-full-run"
$B > $O/nogroup.json 2> $O/nogroup.err
PORT=29611
for cfg in "native 1 1 sharded" "native 2 2 sharded" "native 4 4 sharded" "native 1 1 allreduce" "torch 1 1 sharded"; do
  set -- $cfg
  PORT=$((PORT+1))
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT NGP_DDP_NATIVE=$([ $1 = native ] && echo 1 || echo 0) NGP_DDP_CHUNKS=$2 NGP_DDP_GROUPS=$3 NGP_DDP_EXCHANGE=$4 \
    timeout 200 $B > $O/pg1_$1_$2_$3_$4.json 2> $O/pg1_$1_$2_$3_$4.err
done
timeout 200 python tools/api_host_breakdown.py > $O/api_host.json 2> $O/api_host.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04b/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" in d:
        print(f.split("/")[-1], "ms/step %.4f" % d["ms_per_step"], {k: d.get(k) for k in ("exchange", "exchange_impl", "exchange_ms", "exposed_exchange_ms", "host_ms_per_step")})
    else:
        print(f.split("/")[-1], json.dumps(d)[:1500])
PY
tail -5 $O/pytest.txt
