#!/bin/bash
# Scratch: round-3 late batch A -- parity of the changed kernels, A/B of the MLP backward waves and the slice owners' write-out,
# K sweep of the two-round forward with the compact first-round list.
cd /root/repo; mkdir -p gpurun_out/r3a; O=gpurun_out/r3a
timeout 600 python -m pytest tests/test_field_gpu.py tests/test_hashgrid_gpu.py "tests/test_vren_gpu.py::test_first_k_lists_of_the_train_write" -x -q -m gpu -k "bwd or backward or first_k or field" > $O/tests1.txt 2>&1; echo "tests1 rc=$?" >> $O/tests1.txt
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "two_round or native or oracle" > $O/tests2.txt 2>&1; echo "tests2 rc=$?" >> $O/tests2.txt
for lib in libngp_hip.so variants/libngp_hip_mlp4.so; do for a in 175000 25000; do NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/$lib python tools/bench_mlp.py 290000 $a 2>&1 | grep -v amdgpu.ids; done; done > $O/ab_mlp.txt 2>&1
for lib in libngp_hip.so variants/libngp_hip_wo0.so variants/libngp_hip_wo2.so; do for a in 155000 20000; do NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/$lib python tools/bench_bwd.py $a 2>&1 | grep -v amdgpu.ids; done; done > $O/ab_bwd.txt 2>&1
for K in 32 64; do echo "== K=$K 8000"; NGP_TWO_ROUND=on NGP_TWO_ROUND_K=$K STEPS=8000 timeout 120 python tools/late_stage_times.py 2>&1 | grep -v amdgpu.ids; done > $O/late.txt 2>&1
for K in 64; do echo "== K=$K 25000"; NGP_TWO_ROUND=on NGP_TWO_ROUND_K=$K STEPS=25000 timeout 120 python tools/late_stage_times.py 2>&1 | grep -v amdgpu.ids; done >> $O/late.txt 2>&1
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
NGP_HIP_LIB=$PWD/ngp_pl_amd/csrc/variants/libngp_hip_mlp4.so timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-api > $O/bench_mlp4.json 2> $O/bench_mlp4.err
python - <<'P'
import json
for n in ("bench","bench_mlp4"):
    try:
        d=json.load(open("gpurun_out/r3a/%s.json"%n)); print(n, round(d["value"]/1e6,2), d["ms_per_step"], [(s["stage"],s["ms"]) for s in d["roofline"]["stages"]])
    except Exception as e: print(n, "ERR", e)
P
tail -3 $O/tests1.txt $O/tests2.txt; cat $O/ab_mlp.txt $O/ab_bwd.txt $O/late.txt
