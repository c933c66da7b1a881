#!/bin/bash
cd /root/repo; O=gpurun_out/r3g; mkdir -p $O
run() { echo "== $*"; env "${@:3}" python tools/profile_fwd_levels.py $1 $2 2>&1 | grep -v amdgpu.ids | grep "all 16"; }
{ for m in pairs balanced; do
    run 1300000 1.0 NGP_FWD_MAP=$m
    run 1300000 2.0 NGP_FWD_MAP=$m
    run 100000 4.0 NGP_FWD_MAP=$m
  done
} > $O/levels.txt 2>&1
for m in pairs balanced; do echo "== $m"; CONFIGS=device_exact,k4_cap64 NGP_FWD_MAP=$m timeout 120 python tools/profile_render.py 2>&1 | grep -v amdgpu.ids | tail -8; done > $O/render.txt 2>&1
cat $O/levels.txt $O/render.txt
