mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
(time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30) > $O/pytest.txt 2>&1
for i in 1 2; do
  (time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err) 2> $O/bench_$i.time
done
timeout 600 bash tools/profile_bench.sh r04final > $O/profile.log 2>&1
timeout 400 bash tools/render_trained_trace.sh r04final > $O/render_trace.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04j/bench_*.json")):
    d = json.load(open(f)); fr = d.get("full_run", {})
    print(f.split("/")[-1], "value %.4g ms/step %.4f vr_s %.2f" % (d["value"], d["ms_per_step"], d["config"]["samples_per_ray_composited"]),
          "full_run %.2f s psnr %.2f fps %.1f / %.1f" % (fr.get("train_s", 0), fr.get("psnr", 0), fr.get("fps_200", 0), fr.get("fps_200_reference_chunking", 0)),
          "api %.4f plain %.4f" % (d["api_path"]["ms_per_step"], d["api_path_plain"]["ms_per_step"]), "roof %.3f" % d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
PY
tail -5 $O/pytest.txt; tail -2 $O/smoke.txt
