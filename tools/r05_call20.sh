#!/bin/bash
# round 5, call 20: the driver's literal command twice more on another box (the step did not change since call 13; box-to-box spread)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c20_run$i.json 2> gpurun_out/r05_c20_run$i.err
  echo "run $i rc=$?"
done
python - <<'P'
import json
for i in (1,2):
    d=json.load(open('gpurun_out/r05_c20_run%d.json'%i))
    print(round(d['value']/1e6,3), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['api_path']['rays_per_s']/1e6,2), round(d['api_path_plain']['rays_per_s']/1e6,2), round(d['api_path_reference_files']['rays_per_s']/1e6,2),
          [round(x['rays_per_s']/1e6,2) for x in d['secondary']], round(d['full_run']['train_s'],3), round(d['full_run']['psnr'],4), round(d['render_fps_800x800']['fps'],1), round(d['render_fps_800x800_regrouped']['fps'],1))
P
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
