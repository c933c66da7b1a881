// TEST INFRASTRUCTURE.  C ABI over the reference's own host entry points (`*_cu`, declared in
// /root/reference/models/csrc/include/utils.h) when its .cu files are compiled for the CPU by
// oracle/build_ref.sh.  Plays the role binding.cpp:234-250 plays for the real extension.
#include "utils.h"
#include <cstring>

thread_local uint3_ threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob((void*)p, shape, torch::kFloat32); }
torch::Tensor i32(const int32_t* p, std::vector<int64_t> shape) { return torch::from_blob((void*)p, shape, torch::kInt32); }
torch::Tensor i64(const int64_t* p, std::vector<int64_t> shape) { return torch::from_blob((void*)p, shape, torch::kInt64); }
torch::Tensor u8(const uint8_t* p, std::vector<int64_t> shape) { return torch::from_blob((void*)p, shape, torch::kUInt8); }
template <typename T> void out(const torch::Tensor& t, T* dst, int64_t n) {
    auto c = t.contiguous();
    std::memcpy(dst, c.data_ptr<T>(), sizeof(T) * (size_t)n);
}
}  // namespace

extern "C" {

void ref_morton3D(const int32_t* coords, int n, int32_t* indices) { out(morton3D_cu(i32(coords, {n, 3})), indices, n); }
void ref_morton3D_invert(const int32_t* indices, int n, int32_t* coords) { out(morton3D_invert_cu(i32(indices, {n})), coords, 3LL * n); }
void ref_packbits(const float* grid, int n_bytes, float thr, uint8_t* bitfield) {
    packbits_cu(f32(grid, {8LL * n_bytes}), thr, u8(bitfield, {n_bytes}));
}

void ref_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* half_sizes,
                            int n_rays, int n_voxels, int max_hits, int32_t* hit_cnt, float* hits_t, int64_t* hits_voxel_idx) {
    auto r = ray_aabb_intersect_cu(f32(rays_o, {n_rays, 3}), f32(rays_d, {n_rays, 3}), f32(centers, {n_voxels, 3}),
                                   f32(half_sizes, {n_voxels, 3}), max_hits);
    out(r[0], hit_cnt, n_rays); out(r[1], hits_t, 2LL * n_rays * max_hits); out(r[2], hits_voxel_idx, (int64_t)n_rays * max_hits);
}
void ref_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* radii,
                              int n_rays, int n_spheres, int max_hits, int32_t* hit_cnt, float* hits_t, int64_t* hits_idx) {
    auto r = ray_sphere_intersect_cu(f32(rays_o, {n_rays, 3}), f32(rays_d, {n_rays, 3}), f32(centers, {n_spheres, 3}),
                                     f32(radii, {n_spheres}), max_hits);
    out(r[0], hit_cnt, n_rays); out(r[1], hits_t, 2LL * n_rays * max_hits); out(r[2], hits_idx, (int64_t)n_rays * max_hits);
}

// returns total samples; outputs are the reference's slices [:total]
long long ref_raymarching_train(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* bitfield,
                                int cascades, float scale, float exp_step_factor, const float* noise, int grid_size,
                                int max_samples, int n_rays, long long cap, long long n_bitfield,
                                int64_t* rays_a, float* xyzs, float* dirs, float* deltas, float* ts, int32_t* counter) {
    auto r = raymarching_train_cu(f32(rays_o, {n_rays, 3}), f32(rays_d, {n_rays, 3}), f32(hits_t, {n_rays, 2}),
                                  u8(bitfield, {n_bitfield}), cascades, scale, exp_step_factor, f32(noise, {n_rays}),
                                  grid_size, max_samples);
    out(r[5], counter, 2);
    const long long S = counter[0];
    if (S > cap) return -1;
    out(r[0], rays_a, 3LL * n_rays);
    out(r[1].slice(0, 0, S), xyzs, 3 * S); out(r[2].slice(0, 0, S), dirs, 3 * S);
    out(r[3].slice(0, 0, S), deltas, S); out(r[4].slice(0, 0, S), ts, S);
    return S;
}

void ref_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive, const uint8_t* bitfield,
                          int cascades, float scale, float exp_step_factor, int grid_size, int max_samples, int n_samples,
                          int n_alive, int n_rays_total, long long n_bitfield,
                          float* xyzs, float* dirs, float* deltas, float* ts, int32_t* n_eff) {
    auto r = raymarching_test_cu(f32(rays_o, {n_rays_total, 3}), f32(rays_d, {n_rays_total, 3}), f32(hits_t, {n_rays_total, 2}),
                                 i64(alive, {n_alive}), u8(bitfield, {n_bitfield}), cascades, scale, exp_step_factor,
                                 grid_size, max_samples, n_samples);
    const int64_t n = (int64_t)n_alive * n_samples;
    out(r[0], xyzs, 3 * n); out(r[1], dirs, 3 * n); out(r[2], deltas, n); out(r[3], ts, n); out(r[4], n_eff, n_alive);
}

void ref_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts, const int64_t* rays_a,
                            float T_threshold, int n_rays, int n_samples,
                            int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws) {
    auto r = composite_train_fw_cu(f32(sigmas, {n_samples}), f32(rgbs, {n_samples, 3}), f32(deltas, {n_samples}),
                                   f32(ts, {n_samples}), i64(rays_a, {n_rays, 3}), T_threshold);
    out(r[0], total_samples, n_rays); out(r[1], opacity, n_rays); out(r[2], depth, n_rays); out(r[3], rgb, 3LL * n_rays); out(r[4], ws, n_samples);
}

void ref_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                            const float* sigmas, const float* rgbs, const float* ws, const float* deltas, const float* ts,
                            const int64_t* rays_a, const float* opacity, const float* depth, const float* rgb,
                            float T_threshold, int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs) {
    auto r = composite_train_bw_cu(f32(dL_dopacity, {n_rays}), f32(dL_ddepth, {n_rays}), f32(dL_drgb, {n_rays, 3}), f32(dL_dws, {n_samples}),
                                   f32(sigmas, {n_samples}), f32(rgbs, {n_samples, 3}), f32(ws, {n_samples}), f32(deltas, {n_samples}),
                                   f32(ts, {n_samples}), i64(rays_a, {n_rays, 3}), f32(opacity, {n_rays}), f32(depth, {n_rays}),
                                   f32(rgb, {n_rays, 3}), T_threshold);
    out(r[0], dL_dsigmas, n_samples); out(r[1], dL_drgbs, 3LL * n_samples);
}

void ref_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts, const float* hits_t,
                           int64_t* alive, float T_threshold, const int32_t* n_eff, int n_alive, int n_samples, int n_rays_total,
                           float* opacity, float* depth, float* rgb) {
    composite_test_fw_cu(f32(sigmas, {n_alive, n_samples}), f32(rgbs, {n_alive, n_samples, 3}), f32(deltas, {n_alive, n_samples}),
                         f32(ts, {n_alive, n_samples}), f32(hits_t, {n_rays_total, 2}), i64(alive, {n_alive}), T_threshold,
                         i32(n_eff, {n_alive}), f32(opacity, {n_rays_total}), f32(depth, {n_rays_total}), f32(rgb, {n_rays_total, 3}));
}

void ref_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int n_rays, int n_samples,
                            float* loss, float* ws_incl, float* wts_incl) {
    auto r = distortion_loss_fw_cu(f32(ws, {n_samples}), f32(deltas, {n_samples}), f32(ts, {n_samples}), i64(rays_a, {n_rays, 3}));
    out(r[0], loss, n_rays); out(r[1], ws_incl, n_samples); out(r[2], wts_incl, n_samples);
}
void ref_distortion_loss_bw(const float* dL_dloss, const float* ws_incl, const float* wts_incl, const float* ws, const float* deltas,
                            const float* ts, const int64_t* rays_a, int n_rays, int n_samples, float* dL_dws) {
    auto r = distortion_loss_bw_cu(f32(dL_dloss, {n_rays}), f32(ws_incl, {n_samples}), f32(wts_incl, {n_samples}), f32(ws, {n_samples}),
                                   f32(deltas, {n_samples}), f32(ts, {n_samples}), i64(rays_a, {n_rays, 3}));
    out(r, dL_dws, n_samples);
}

}  // extern "C"
