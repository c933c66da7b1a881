#!/bin/bash
# round 5, call 21: did the late-round march.hip slow the API-shaped legs?  The product library against a build with the march.hip of
# before the frame-loop work (ngp_pl_amd/csrc/variants/libngp_hip_march_old.so), same box, alternating, three rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2 3; do
  for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/libngp_hip_march_old.so; do
    NGP_HIP_LIB=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-secondary --no-full-run 2>/dev/null |
      python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$rep %-22s value %.2f M  api_path %.2f  api_path_plain %.2f  reference_files %.2f' % ('$lib'.split('/')[-1][:22], d['value']/1e6, d['api_path']['rays_per_s']/1e6, d['api_path_plain']['rays_per_s']/1e6, d['api_path_reference_files']['rays_per_s']/1e6))"
  done
done
