cd /root/repo
for lib in ngp_pl_amd/csrc/variants/libngp_hip_base.so ngp_pl_amd/csrc/libngp_hip.so ngp_pl_amd/csrc/variants/libngp_hip_new_b8.so ngp_pl_amd/csrc/variants/libngp_hip_new_B_s4608_x2_b5.so ngp_pl_amd/csrc/variants/libngp_hip_base_timing.so ngp_pl_amd/csrc/variants/libngp_hip_timing.so; do
  NGP_HIP_LIB=$PWD/$lib python tools/bench_bwd.py 155000 2>&1 | tail -3
done
