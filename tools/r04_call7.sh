mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run --no-api"
run() { name=$1; shift; env "$@" timeout 120 $B > $O/$name.json 2> $O/$name.err; }
run default A=1
run at_hashgrid_bwd NGP_MARCH_AT=hashgrid_bwd
run at_adam NGP_MARCH_AT=adam
run prio0 NGP_MARCH_PRIORITY=0
run prio1 NGP_MARCH_PRIORITY=1
run at_hashgrid_bwd_prio0 NGP_MARCH_AT=hashgrid_bwd NGP_MARCH_PRIORITY=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-300:]); continue
    r = d.get("roofline", {})
    st = {s["stage"]: s["ms"] for s in r.get("stages", [])}
    print("%-24s ms/step %.4f vr_s %.3f stage_sum %.4f | write %.4f hfwd %.4f mlpf %.4f cfw %.4f cbw %.4f mlpb %.4f hbwd %.4f adam %.4f march %.4f host_wait %.3f" % (f.split("/")[-1][:-5], d["ms_per_step"],
          d["config"]["samples_per_ray_composited"], r.get("main_stream_stage_sum_ms", 0), st.get("march_write", 0), st.get("hashgrid_fwd", 0), st.get("mlp_fwd", 0), st.get("composite_fw+loss", 0), st.get("composite_bw", 0), st.get("mlp_bwd", 0), st.get("hashgrid_bwd", 0), st.get("adam", 0), st.get("march_count(side stream)", 0), d["host_ms_per_step"]["wait_ms_per_step"]))
PY
