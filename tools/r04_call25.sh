#!/bin/bash
# Round 4, call 25: the whole GPU suite, smoke, and the driver's command three times on the late-round tree.
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/*.json
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25) > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for i in 1 2 3; do timeout 280 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$i.json 2> $O/bench$i.err; done
python - <<'PY'
import json
for f in ("bench1", "bench2", "bench3"):
    try:
        b = json.loads(open("gpurun_out/r04z/%s.json" % f).read().strip().splitlines()[-1])
        fr = b.get("full_run") or {}
        rf = b["roofline"]
        print(f, "value %.4e" % b["value"], "ms/step %.4f" % b["ms_per_step"], "frac %.3f" % rf["frac"], rf["kernel"], "main sum", rf.get("main_stream_stage_sum_ms"),
              "| api %.4f plain %.4f" % (b["api_path"]["ms_per_step"], b["api_path_plain"]["ms_per_step"]),
              "| full_run", {k: (round(fr[k], 3) if isinstance(fr.get(k), float) else fr.get(k)) for k in ("train_s", "psnr", "fps_200", "fps_200_reference_chunking")},
              "| cpu", (b.get("cpu_baseline") or {}).get("value"))
        print("   stages", [(d["stage"], d["ms"]) for d in rf["stages"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
