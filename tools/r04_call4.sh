mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
(time timeout 600 python -m pytest "tests/test_train_gpu.py::test_training_is_reproducible_run_to_run" "tests/test_train_gpu.py::test_native_occupancy_update_matches_reference_semantics" "tests/test_field_gpu.py" -q -p no:cacheprovider -k "reproducible or occupancy or hashgrid_backward" 2>&1 | tail -15) > $O/pytest.txt 2>&1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run --no-api"
V=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_nopf.so
for i in 1 2 3; do
  NGP_HIP_LIB=$V $B > $O/nopf_$i.json 2> $O/nopf_$i.err
  $B > $O/pf_$i.json 2> $O/pf_$i.err
  NGP_FUSED_TAIL=0 $B > $O/pf_tail0_$i.json 2> $O/pf_tail0_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04d/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline", {})
    st = {s["stage"]: s["ms"] for s in r.get("stages", [])}
    print("%-18s ms/step %.4f (win %.4f-%.4f) vr_s %.3f active %.0f stage_sum %.4f bwd %.4f cfw %.4f cbw %.4f mlpb %.4f" % (f.split("/")[-1], d["ms_per_step"], d["window_ms_per_step_min_max"][0], d["window_ms_per_step_min_max"][1],
          d["config"]["samples_per_ray_composited"], r.get("samples_active_per_launch", 0), r.get("main_stream_stage_sum_ms", 0), st.get("hashgrid_bwd", 0), st.get("composite_fw+loss", 0), st.get("composite_bw", 0), st.get("mlp_bwd", 0)))
PY
tail -4 $O/pytest.txt
