mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
(time timeout 900 python -m pytest tests/test_properties_gpu.py "tests/test_train_gpu.py::test_two_round_forward_ignores_stale_values_behind_a_stop" "tests/test_train_gpu.py::test_native_stepper_equals_the_python_enqueue_path" "tests/test_train_gpu.py::test_two_round_forward_is_bit_identical" tests/test_reference_surface_gpu.py "tests/test_train_gpu.py::test_fused_step_equals_autograd_step" -q -p no:cacheprovider 2>&1 | tail -30) > $O/pytest.txt 2>&1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run"
for i in 1 2; do
  NGP_FUSED_TAIL=0 $B > $O/tail0_$i.json 2> $O/tail0_$i.err
  NGP_FUSED_TAIL=1 $B > $O/tail1_$i.json 2> $O/tail1_$i.err
done
timeout 200 python tools/api_host_breakdown.py > $O/api_host.json 2> $O/api_host.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04c/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" in d:
        r = d.get("roofline", {})
        print(f.split("/")[-1], "ms/step %.4f" % d["ms_per_step"], "api %.4f plain %.4f" % (d["api_path"]["ms_per_step"], d["api_path_plain"]["ms_per_step"]),
              "stage sum", r.get("main_stream_stage_sum_ms"), [(s["stage"], s["ms"]) for s in r.get("stages", [])])
    else:
        print(f.split("/")[-1], json.dumps(d)[:1200])
PY
tail -8 $O/pytest.txt
