#!/bin/bash
cd /root/repo; O=gpurun_out/r3i; mkdir -p $O
timeout 400 python -m pytest tests/test_field_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
python tools/profile_fwd_levels.py 2>&1 | grep -v amdgpu.ids | grep "all 16\|level  0\|level  8\|level 15" > $O/levels.txt
python tools/profile_fwd_levels.py 1300000 2.0 2>&1 | grep -v amdgpu.ids | grep "all 16" >> $O/levels.txt
timeout 250 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['api_path']['rays_per_s'], d['render_fps_800x800']['fps'], d['render_fps_800x800_reference_chunking']['fps'], d['roofline']['frac'], [(s['stage'],s['ms']) for s in d['roofline']['stages'] if 'fwd' in s['stage']])"
tail -n 3 $O/tests.txt; cat $O/levels.txt
