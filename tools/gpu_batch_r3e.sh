#!/bin/bash
# Scratch: round-3 batch E -- hash forward: cell-run reuse x cost-weighted XCD map.
cd /root/repo; O=gpurun_out/r3e; mkdir -p $O
NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=weighted timeout 400 python -m pytest tests/test_field_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
run() { echo "== $*"; env "$@" python tools/profile_fwd_levels.py 2>&1 | grep -v amdgpu.ids | grep "all 16\|level  0\|level 15\|level  8"; }
{ run NGP_FWD_REUSE_MAX_RES=0 NGP_FWD_MAP=pairs
  run NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=pairs
  run NGP_FWD_REUSE_MAX_RES=0 NGP_FWD_MAP=weighted
  run NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=weighted NGP_FWD_WEIGHT_BASE=0
  run NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=weighted NGP_FWD_WEIGHT_BASE=5
  run NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=weighted NGP_FWD_WEIGHT_BASE=15
  run NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=weighted NGP_FWD_WEIGHT_BASE=40
  run NGP_FWD_REUSE_MAX_RES=300 NGP_FWD_MAP=weighted NGP_FWD_WEIGHT_BASE=5
} > $O/levels.txt
for cfg in "NGP_FWD_REUSE_MAX_RES=0 NGP_FWD_MAP=pairs" "NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=pairs" "NGP_FWD_REUSE_MAX_RES=4096 NGP_FWD_MAP=weighted"; do
  echo "== $cfg"; env $cfg timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], d['render_fps_800x800']['fps'], d['render_fps_800x800_reference_chunking']['fps'], [(s['stage'],s['ms']) for s in d['roofline']['stages'] if 'fwd' in s['stage']], d['config']['train_psnr'])"
done > $O/bench.txt 2>&1
tail -n 3 $O/tests.txt; cat $O/levels.txt $O/bench.txt
