// Volumetric compositing (train fwd/bwd, test fwd) and distortion loss for gfx950.
//
// Semantics: /root/reference/models/csrc/volumerendering.cu and losses.cu (cited per kernel).
// Arithmetic order inside a ray is the reference's front-to-back order (no contraction), so
// the only divergence from the CPU oracle is the fast exponential (__expf, as the reference).
// Differences in structure: outputs are fully written by the kernels (no host-side zero fill),
// the backward's prefix sum of dL/dw*w runs in registers in the same pass order instead of an
// in-thread thrust::inclusive_scan over a pre-multiplied global buffer.
#pragma clang fp contract(off)

#include "ngp_common.h"

namespace {

// volumerendering.cu:20-44
__global__ void __launch_bounds__(64)
composite_train_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                          const float* __restrict__ deltas, const float* __restrict__ ts,
                          const int64_t* __restrict__ rays_a, float T_threshold, int n_rays,
                          int64_t* __restrict__ total_samples, float* __restrict__ opacity,
                          float* __restrict__ depth, float* __restrict__ rgb, float* __restrict__ ws) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * (size_t)n];
    const int64_t start = rays_a[3 * (size_t)n + 1];
    const int N = (int)rays_a[3 * (size_t)n + 2];
    int samples = 0;
    float T = 1.0f, R = 0.f, G = 0.f, B = 0.f, D = 0.f, O = 0.f;
    int k = 0;
    for (; k < N; ++k) {
        const size_t s = (size_t)start + k;
        const float a = 1.0f - __expf(-sigmas[s] * deltas[s]);
        const float w = a * T;
        R += w * rgbs[3 * s]; G += w * rgbs[3 * s + 1]; B += w * rgbs[3 * s + 2];
        D += w * ts[s];
        O += w;
        ws[s] = w;
        T *= 1.0f - a;
        if (T <= T_threshold) { ++k; break; }
        ++samples;
    }
    for (; k < N; ++k) ws[(size_t)start + k] = 0.0f;   // samples past the stop keep w = 0
    rgb[3 * ray_idx] = R; rgb[3 * ray_idx + 1] = G; rgb[3 * ray_idx + 2] = B;
    depth[ray_idx] = D; opacity[ray_idx] = O;
    total_samples[ray_idx] = samples;
}

// volumerendering.cu:106-150 (+ host pre-multiply :175)
__global__ void __launch_bounds__(64)
composite_train_bw_kernel(const float* __restrict__ dL_dopacity, const float* __restrict__ dL_ddepth,
                          const float* __restrict__ dL_drgb, const float* __restrict__ dL_dws,
                          const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                          const float* __restrict__ ws, const float* __restrict__ deltas,
                          const float* __restrict__ ts, const int64_t* __restrict__ rays_a,
                          const float* __restrict__ opacity, const float* __restrict__ depth,
                          const float* __restrict__ rgb, float T_threshold, int n_rays,
                          float* __restrict__ dL_dsigmas, float* __restrict__ dL_drgbs) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * (size_t)n];
    const int64_t start = rays_a[3 * (size_t)n + 1];
    const int N = (int)rays_a[3 * (size_t)n + 2];
    const float R = rgb[3 * ray_idx], G = rgb[3 * ray_idx + 1], B = rgb[3 * ray_idx + 2];
    const float O = opacity[ray_idx], D = depth[ray_idx];
    const float gR = dL_drgb[3 * ray_idx], gG = dL_drgb[3 * ray_idx + 1], gB = dL_drgb[3 * ray_idx + 2];
    const float gO = dL_dopacity[ray_idx], gD = dL_ddepth[ray_idx];
    // total of dL/dw * w over the whole segment, summed front to back like the inclusive scan
    float P_total = 0.f;
    if (dL_dws != nullptr)
        for (int k = 0; k < N; ++k) { const size_t s = (size_t)start + k; P_total += dL_dws[s] * ws[s]; }
    float T = 1.0f, r = 0.f, g = 0.f, b = 0.f, d = 0.f, P = 0.f;
    int k = 0;
    for (; k < N; ++k) {
        const size_t s = (size_t)start + k;
        const float a = 1.0f - __expf(-sigmas[s] * deltas[s]);
        const float w = a * T;
        const float cr = rgbs[3 * s], cg = rgbs[3 * s + 1], cb = rgbs[3 * s + 2];
        const float tk = ts[s];
        const float gw = dL_dws ? dL_dws[s] : 0.f;
        r += w * cr; g += w * cg; b += w * cb;
        d += w * tk;
        T *= 1.0f - a;
        if (dL_dws) P += gw * ws[s];
        dL_drgbs[3 * s] = gR * w; dL_drgbs[3 * s + 1] = gG * w; dL_drgbs[3 * s + 2] = gB * w;
        dL_dsigmas[s] = deltas[s] * (
            gR * (cr * T - (R - r)) +
            gG * (cg * T - (G - g)) +
            gB * (cb * T - (B - b)) +
            gO * (1 - O) +
            gD * (tk * T - (D - d)) +
            T * gw - (P_total - P));
        if (T <= T_threshold) { ++k; break; }
    }
    for (; k < N; ++k) {
        const size_t s = (size_t)start + k;
        dL_dsigmas[s] = 0.f; dL_drgbs[3 * s] = 0.f; dL_drgbs[3 * s + 1] = 0.f; dL_drgbs[3 * s + 2] = 0.f;
    }
}

// volumerendering.cu:219-248
__global__ void __launch_bounds__(64)
composite_test_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                         const float* __restrict__ deltas, const float* __restrict__ ts,
                         int64_t* __restrict__ alive, float T_threshold,
                         const int32_t* __restrict__ n_eff, int n_alive, int n_samples,
                         float* __restrict__ opacity, float* __restrict__ depth, float* __restrict__ rgb) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int N = n_eff[n];
    if (N == 0) { alive[n] = -1; return; }
    const size_t r = (size_t)alive[n];
    const size_t base = (size_t)n * n_samples;
    float O = opacity[r], D = depth[r], R = rgb[3 * r], G = rgb[3 * r + 1], B = rgb[3 * r + 2];
    float T = 1 - O;
    for (int s = 0; s < N; ++s) {
        const size_t o = base + s;
        const float a = 1.0f - __expf(-sigmas[o] * deltas[o]);
        const float w = a * T;
        R += w * rgbs[3 * o]; G += w * rgbs[3 * o + 1]; B += w * rgbs[3 * o + 2];
        D += w * ts[o];
        O += w;
        T *= 1.0f - a;
        if (T <= T_threshold) { alive[n] = -1; break; }
    }
    opacity[r] = O; depth[r] = D; rgb[3 * r] = R; rgb[3 * r + 1] = G; rgb[3 * r + 2] = B;
}

// losses.cu:9-61 + the elementwise formula :94-95, one pass, scans kept in registers
__global__ void __launch_bounds__(64)
distortion_fw_kernel(const float* __restrict__ ws, const float* __restrict__ deltas,
                     const float* __restrict__ ts, const int64_t* __restrict__ rays_a, int n_rays,
                     float* __restrict__ loss, float* __restrict__ ws_incl, float* __restrict__ wts_incl) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * (size_t)n];
    const int64_t start = rays_a[3 * (size_t)n + 1];
    const int N = (int)rays_a[3 * (size_t)n + 2];
    float w_ex = 0.f, wt_ex = 0.f, acc = 0.f;
    for (int k = 0; k < N; ++k) {
        const size_t s = (size_t)start + k;
        const float w = ws[s], wt = w * ts[s];
        const float w_in = w_ex + w, wt_in = wt_ex + wt;
        ws_incl[s] = w_in; wts_incl[s] = wt_in;
        // 2*(wts_incl*ws_excl - ws_incl*wts_excl) + 1/3*ws*ws*deltas, ATen op order
        const float l = 2 * (wt_in * w_ex - w_in * wt_ex) + (1.0f / 3) * w * w * deltas[s];
        acc += l;
        w_ex = w_in; wt_ex = wt_in;
    }
    loss[ray_idx] = acc;
}

// losses.cu:119-141
__global__ void __launch_bounds__(64)
distortion_bw_kernel(const float* __restrict__ dL_dloss, const float* __restrict__ ws_incl,
                     const float* __restrict__ wts_incl, const float* __restrict__ ws,
                     const float* __restrict__ deltas, const float* __restrict__ ts,
                     const int64_t* __restrict__ rays_a, int n_rays, float* __restrict__ dL_dws) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * (size_t)n];
    const int64_t start = rays_a[3 * (size_t)n + 1];
    const int N = (int)rays_a[3 * (size_t)n + 2];
    if (N <= 0) return;
    const size_t end = (size_t)start + N - 1;
    const float ws_sum = ws_incl[end], wts_sum = wts_incl[end];
    const float gl = dL_dloss[ray_idx];
    float w_prev = 0.f, wt_prev = 0.f;
    for (size_t s = (size_t)start; s <= end; ++s) {
        const float w_in = ws_incl[s], wt_in = wts_incl[s];
        const float t = ts[s];
        float v = gl * 2 * ((s == (size_t)start ? 0.f : (t * w_prev - wt_prev)) +
                            (wts_sum - wt_in - t * (ws_sum - w_in)));
        v += gl * (2.0f / 3) * ws[s] * deltas[s];
        dL_dws[s] = v;
        w_prev = w_in; wt_prev = wt_in;
    }
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int ngp_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                           const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                           int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                           ngp_stream_t stream) {
    if (n_rays < 0 || n_samples < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(total_samples); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    if (n_samples > 0) { NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(ws); }
    hipLaunchKernelGGL(composite_train_fw_kernel, dim3(ngp_div_up(n_rays, 64)), dim3(64), 0, ngp_stream(stream),
                       sigmas, rgbs, deltas, ts, rays_a, T_threshold, n_rays, total_samples, opacity, depth, rgb, ws);
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                           const float* dL_dws, const float* sigmas, const float* rgbs, const float* ws,
                           const float* deltas, const float* ts, const int64_t* rays_a,
                           const float* opacity, const float* depth, const float* rgb, float T_threshold,
                           int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs, ngp_stream_t stream) {
    if (n_rays < 0 || n_samples < 0) return NGP_EINVAL;
    if (n_rays == 0 || n_samples == 0) return 0;
    NGP_CHECK_PTR(dL_dopacity); NGP_CHECK_PTR(dL_ddepth); NGP_CHECK_PTR(dL_drgb); NGP_CHECK_PTR(sigmas);
    NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(ws); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(rays_a);
    NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(dL_dsigmas); NGP_CHECK_PTR(dL_drgbs);
    hipLaunchKernelGGL(composite_train_bw_kernel, dim3(ngp_div_up(n_rays, 64)), dim3(64), 0, ngp_stream(stream),
                       dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a,
                       opacity, depth, rgb, T_threshold, n_rays, dL_dsigmas, dL_drgbs);
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                          int64_t* alive_indices, float T_threshold, const int32_t* n_eff_samples,
                          int n_alive, int n_samples, float* opacity, float* depth, float* rgb,
                          ngp_stream_t stream) {
    if (n_alive < 0 || n_samples < 1) return NGP_EINVAL;
    if (n_alive == 0) return 0;
    NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(alive_indices);
    NGP_CHECK_PTR(n_eff_samples); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    hipLaunchKernelGGL(composite_test_fw_kernel, dim3(ngp_div_up(n_alive, 64)), dim3(64), 0, ngp_stream(stream),
                       sigmas, rgbs, deltas, ts, alive_indices, T_threshold, n_eff_samples, n_alive, n_samples,
                       opacity, depth, rgb);
    return NGP_LAUNCH_RESULT();
}

int ngp_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a,
                           int n_rays, int n_samples, float* loss, float* ws_inclusive_scan,
                           float* wts_inclusive_scan, ngp_stream_t stream) {
    if (n_rays < 0 || n_samples < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(loss);
    if (n_samples > 0) { NGP_CHECK_PTR(ws); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(ws_inclusive_scan); NGP_CHECK_PTR(wts_inclusive_scan); }
    hipLaunchKernelGGL(distortion_fw_kernel, dim3(ngp_div_up(n_rays, 64)), dim3(64), 0, ngp_stream(stream),
                       ws, deltas, ts, rays_a, n_rays, loss, ws_inclusive_scan, wts_inclusive_scan);
    return NGP_LAUNCH_RESULT();
}

int ngp_distortion_loss_bw(const float* dL_dloss, const float* ws_inclusive_scan, const float* wts_inclusive_scan,
                           const float* ws, const float* deltas, const float* ts, const int64_t* rays_a,
                           int n_rays, int n_samples, float* dL_dws, ngp_stream_t stream) {
    if (n_rays < 0 || n_samples < 0) return NGP_EINVAL;
    if (n_rays == 0 || n_samples == 0) return 0;
    NGP_CHECK_PTR(dL_dloss); NGP_CHECK_PTR(ws_inclusive_scan); NGP_CHECK_PTR(wts_inclusive_scan); NGP_CHECK_PTR(ws);
    NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(dL_dws);
    hipLaunchKernelGGL(distortion_bw_kernel, dim3(ngp_div_up(n_rays, 64)), dim3(64), 0, ngp_stream(stream),
                       dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a, n_rays, dL_dws);
    return NGP_LAUNCH_RESULT();
}

}  // extern "C"
