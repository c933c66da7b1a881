#!/bin/bash
# Scratch: round-3 batch F -- hash forward: cost-balanced XCD map (whole large tables + dense levels in sixteenths).
cd /root/repo; O=gpurun_out/r3f; mkdir -p $O
timeout 400 python -m pytest tests/test_field_gpu.py -x -q -m gpu -k "hashgrid_forward or field_forward or lds_resident or outside" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
run() { echo "== $*"; env "$@" python tools/profile_fwd_levels.py 2>&1 | grep -v amdgpu.ids | grep "all 16"; }
{ run NGP_FWD_MAP=pairs
  run NGP_FWD_MAP=balanced
  run NGP_FWD_MAP=balanced NGP_FWD_SMALL_COST=8
  run NGP_FWD_MAP=balanced NGP_FWD_SMALL_COST=24
  run NGP_FWD_MAP=balanced NGP_FWD_SMALL_COST=4
} > $O/levels.txt
for cfg in "NGP_FWD_MAP=pairs" "NGP_FWD_MAP=balanced"; do
  echo "== $cfg"; env $cfg timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], d['render_fps_800x800']['fps'], d['render_fps_800x800_reference_chunking']['fps'], [(s['stage'],s['ms']) for s in d['roofline']['stages'] if 'fwd' in s['stage']], d['config']['train_psnr'])"
done > $O/bench.txt 2>&1
tail -n 3 $O/tests.txt; cat $O/levels.txt $O/bench.txt
