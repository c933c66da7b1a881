mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
(time timeout 600 python -m pytest "tests/test_train_gpu.py::test_merge_folded_into_adam_is_bit_identical" "tests/test_properties_gpu.py" "tests/test_train_gpu.py::test_native_stepper_equals_the_python_enqueue_path" -q -p no:cacheprovider 2>&1 | tail -12) > $O/pytest.txt 2>&1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run --no-api"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; }
run default A=1
run default2 A=1
run merge0 NGP_MERGE_IN_ADAM=0
run tail0 NGP_FUSED_TAIL=0
run tail0_merge0 NGP_FUSED_TAIL=0 NGP_MERGE_IN_ADAM=0
for at in top hashgrid_fwd composite_fw composite_bw mlp_bwd; do run at_$at NGP_MARCH_AT=$at; done
run at_hashgrid_fwd_tail0 NGP_MARCH_AT=hashgrid_fwd NGP_FUSED_TAIL=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline", {})
    st = {s["stage"]: s["ms"] for s in r.get("stages", [])}
    print("%-24s ms/step %.4f (win %.4f-%.4f) vr_s %.3f stage_sum %.4f | hfwd %.4f mlpf %.4f cfw %.4f cbw %.4f mlpb %.4f hbwd %.4f adam %.4f march %.4f" % (f.split("/")[-1][:-5], d["ms_per_step"], d["window_ms_per_step_min_max"][0], d["window_ms_per_step_min_max"][1],
          d["config"]["samples_per_ray_composited"], r.get("main_stream_stage_sum_ms", 0), st.get("hashgrid_fwd", 0), st.get("mlp_fwd", 0), st.get("composite_fw+loss", 0), st.get("composite_bw", 0), st.get("mlp_bwd", 0), st.get("hashgrid_bwd", 0), st.get("adam", 0), st.get("march_count(side stream)", 0)))
PY
tail -4 $O/pytest.txt
