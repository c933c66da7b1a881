// Optimizer / AMP plumbing / loss seeds for gfx950: dense, HBM-bound streaming kernels.
//
//  * ngp_adam_step    apex FusedAdam semantics (train.py:131: lr 1e-2, eps 1e-15, adam_w_mode,
//                     bias correction) fused with gradient unscale, the f32->f16 parameter cast
//                     tiny-cuda-nn performs each forward, and gradient zeroing: ONE pass over
//                     the 11.4 M parameters (30 B/param with f16 grads) instead of four.
//  * ngp_nerf_loss    NeRFLoss (losses.py:47-60) + mean reduction (train.py:173) + background
//                     blend (rendering.py:153-161) with analytic backward seeds.
#include "ngp_common.h"
#include <hip/hip_fp16.h>

namespace {

typedef _Float16 h1;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

template <bool GRAD_F32>
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ param, h1* __restrict__ param_h, void* __restrict__ grad,
            float* __restrict__ m, float* __restrict__ v, long long n4, long long n,
            float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
            float inv_scale, const int32_t* __restrict__ found_inf) {
    const bool skip = found_inf != nullptr && *found_inf != 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const long long base = q * 4;
        float g[4];
        const int cnt = (int)((n - base) < 4 ? (n - base) : 4);
        if (cnt == 4) {
            if (GRAD_F32) {
                float4* gp = reinterpret_cast<float4*>(reinterpret_cast<float*>(grad) + base);
                const float4 t = *gp; g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
                *gp = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                half4_t* gp = reinterpret_cast<half4_t*>(reinterpret_cast<h1*>(grad) + base);
                const half4_t t = *gp; g[0] = (float)t[0]; g[1] = (float)t[1]; g[2] = (float)t[2]; g[3] = (float)t[3];
                const half4_t z = {0, 0, 0, 0}; *gp = z;
            }
            if (skip) continue;
            float4 p = *reinterpret_cast<float4*>(param + base);
            float4 mm = *reinterpret_cast<float4*>(m + base);
            float4 vv = *reinterpret_cast<float4*>(v + base);
            float* pp = &p.x; float* mp = &mm.x; float* vp = &vv.x;
            half4_t ph;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = g[k] * inv_scale;
                mp[k] = beta1 * mp[k] + (1.f - beta1) * gk;
                vp[k] = beta2 * vp[k] + (1.f - beta2) * gk * gk;
                const float denom = sqrtf(vp[k] / bc2) + eps;
                pp[k] = pp[k] - lr * ((mp[k] / bc1) / denom + wd * pp[k]);
                ph[k] = (h1)pp[k];
            }
            *reinterpret_cast<float4*>(param + base) = p;
            *reinterpret_cast<float4*>(m + base) = mm;
            *reinterpret_cast<float4*>(v + base) = vv;
            if (param_h) *reinterpret_cast<half4_t*>(param_h + base) = ph;
        } else {
            for (int k = 0; k < cnt; ++k) {
                const long long i = base + k;
                float gk;
                if (GRAD_F32) { float* gp = reinterpret_cast<float*>(grad) + i; gk = *gp; *gp = 0.f; }
                else { h1* gp = reinterpret_cast<h1*>(grad) + i; gk = (float)*gp; *gp = (h1)0; }
                if (skip) continue;
                gk *= inv_scale;
                const float mk = beta1 * m[i] + (1.f - beta1) * gk;
                const float vk = beta2 * v[i] + (1.f - beta2) * gk * gk;
                const float denom = sqrtf(vk / bc2) + eps;
                const float pk = param[i] - lr * ((mk / bc1) / denom + wd * param[i]);
                m[i] = mk; v[i] = vk; param[i] = pk;
                if (param_h) param_h[i] = (h1)pk;
            }
        }
    }
}

// out[i] = sum_p partials[p][i].  64 columns per workgroup, 4 row lanes per column: every thread
// keeps n_partials/4 independent loads in flight (a thread per column walking all rows serially
// measured 62 us for 256 x 10240 on MI355X; this is ~5).
__global__ void __launch_bounds__(256)
reduce_partials_kernel(const float* __restrict__ partials, int n_partials, int n, float* __restrict__ out) {
    __shared__ float s_acc[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), lane_row = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < n)
        for (int p = lane_row; p < n_partials; p += 4) acc += partials[(size_t)p * n + col];
    s_acc[lane_row][threadIdx.x & 63] = acc;
    __syncthreads();
    if (lane_row == 0 && col < n) out[col] = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256)
cast_f32_f16_kernel(const float* __restrict__ in, long long n, h1* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (h1)in[i];
}
__global__ void __launch_bounds__(256)
cast_f16_f32_kernel(const h1* __restrict__ in, long long n, float scale, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i] * scale;
}

__global__ void __launch_bounds__(256)
nerf_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ opacity, const float* __restrict__ gt,
                 const float* __restrict__ bg, float lambda_o, float grad_scale, int n_rays,
                 float* __restrict__ loss, float* __restrict__ sq_err,
                 float* __restrict__ dL_drgb, float* __restrict__ dL_dopacity) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f, se = 0.f;
    if (r < n_rays) {
        const float o = opacity[r];
        const float inv_r = 1.0f / (float)n_rays, inv_3r = 1.0f / (3.0f * (float)n_rays);
        float go = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float b = bg ? bg[c] : 0.f;
            const float diff = rgb[3 * r + c] + b * (1.0f - o) - gt[3 * r + c];
            se += diff * diff;
            const float g = 2.0f * diff * inv_3r;
            dL_drgb[3 * r + c] = g * grad_scale;
            go -= g * b;
        }
        const float oe = o + 1e-10f;
        const float lg = __logf(oe);
        l = se * inv_3r + lambda_o * (-oe * lg) * inv_r;
        go += lambda_o * (-(lg + 1.0f)) * inv_r;
        dL_dopacity[r] = go * grad_scale;
    }
    l = ngp_wave_sum(l); se = ngp_wave_sum(se);
    __shared__ float s_l[4], s_e[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_l[w] = l; s_e[w] = se; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss, s_l[0] + s_l[1] + s_l[2] + s_l[3]);
        if (sq_err) atomicAdd(sq_err, s_e[0] + s_e[1] + s_e[2] + s_e[3]);
    }
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int ngp_adam_step(float* param, ngp_half* param_h, void* grad, int grad_is_f32, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  const int32_t* found_inf, ngp_stream_t stream) {
    if (n < 0 || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(param); NGP_CHECK_PTR(grad); NGP_CHECK_PTR(m); NGP_CHECK_PTR(v);
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    const long long n4 = (n + 3) / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    if (grad_is_f32)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, ngp_stream(stream), param, (h1*)param_h, grad, m, v,
                           n4, (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, 1.0f / grad_scale, found_inf);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, ngp_stream(stream), param, (h1*)param_h, grad, m, v,
                           n4, (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, 1.0f / grad_scale, found_inf);
    return NGP_LAUNCH_RESULT();
}

int ngp_reduce_partials(const float* partials, int n_partials, int n, float* out, ngp_stream_t stream) {
    if (n_partials < 0 || n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(out);
    if (n_partials > 0) NGP_CHECK_PTR(partials);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ngp_div_up(n, 64)), dim3(256), 0, ngp_stream(stream), partials, n_partials, n, out);
    return NGP_LAUNCH_RESULT();
}

int ngp_cast_f32_to_f16(const float* in, int64_t n, ngp_half* out, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(in); NGP_CHECK_PTR(out);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(blocks), dim3(256), 0, ngp_stream(stream), in, (long long)n, (h1*)out);
    return NGP_LAUNCH_RESULT();
}

int ngp_cast_f16_to_f32(const ngp_half* in, int64_t n, float scale, float* out, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(in); NGP_CHECK_PTR(out);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(blocks), dim3(256), 0, ngp_stream(stream), (const h1*)in, (long long)n, scale, out);
    return NGP_LAUNCH_RESULT();
}

int ngp_nerf_loss(const float* rgb, const float* opacity, const float* gt_rgb, const float* bg, float lambda_opacity,
                  float grad_scale, int n_rays, float* loss, float* sq_err, float* dL_drgb, float* dL_dopacity,
                  ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(gt_rgb); NGP_CHECK_PTR(loss); NGP_CHECK_PTR(dL_drgb); NGP_CHECK_PTR(dL_dopacity);
    hipLaunchKernelGGL(nerf_loss_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                       rgb, opacity, gt_rgb, bg, lambda_opacity, grad_scale, n_rays, loss, sq_err, dL_drgb, dL_dopacity);
    return NGP_LAUNCH_RESULT();
}

}  // extern "C"
