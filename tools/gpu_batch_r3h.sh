#!/bin/bash
cd /root/repo; O=gpurun_out/r3h; mkdir -p $O
timeout 400 python -m pytest tests/test_field_gpu.py -x -q -m gpu -k "hashgrid_forward or field_forward or lds_resident or outside" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
run() { echo "== $*"; env "${@:3}" python tools/profile_fwd_levels.py $1 $2 2>&1 | grep -v amdgpu.ids | grep "all 16"; }
{ for m in pairs balanced; do
    run 305000 1.0 NGP_FWD_MAP=$m
    run 1300000 2.0 NGP_FWD_MAP=$m
    run 100000 4.0 NGP_FWD_MAP=$m
  done
} > $O/levels.txt 2>&1
for cfg in "NGP_FWD_MAP=pairs" "NGP_FWD_MAP=balanced"; do
  echo "== $cfg"; env $cfg timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], d['render_fps_800x800']['fps'], d['render_fps_800x800_reference_chunking']['fps'], [(s['stage'],s['ms']) for s in d['roofline']['stages'] if 'fwd' in s['stage']], d['config']['train_psnr'])"
done > $O/bench.txt 2>&1
tail -n 2 $O/tests.txt; cat $O/levels.txt $O/bench.txt
