#!/bin/bash
# round 5, call 13: the driver's literal command three times back to back on one box (run-to-run spread of every leg)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c13_run$i.json 2> gpurun_out/r05_c13_run$i.err
  echo "run $i rc=$?"
done
python - <<'P'
import json
runs=[json.load(open('gpurun_out/r05_c13_run%d.json'%i)) for i in (1,2,3)]
out={"command":"python bench.py --gpus 1 --steps 20 --warmup 5 (three runs back to back, one box)", "runs":runs}
json.dump(out, open('gpurun_out/r05_c13_x3.json','w'))
for d in runs:
    print(round(d['value']/1e6,3), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['api_path']['rays_per_s']/1e6,2), round(d['api_path_plain']['rays_per_s']/1e6,2), round(d['api_path_reference_files']['rays_per_s']/1e6,2),
          [round(x['rays_per_s']/1e6,2) for x in d['secondary']], round(d['full_run']['train_s'],3), round(d['full_run']['psnr'],4), round(d['render_fps_800x800']['fps'],1), round(d['render_fps_800x800_regrouped']['fps'],1),
          [round(p['rays_per_s']/1e6,2) for p in d['sensitivity']['points']])
P
