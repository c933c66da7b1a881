cd /root/repo
for lib in ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/*.so; do
  NGP_HIP_LIB=$PWD/$lib python tools/bench_bwd.py 155000 2>&1 | grep -v amdgpu.ids
done
