// Per-ray NeRF loss terms shared by ngp_nerf_loss (optim.hip) and the fused composite + loss kernel
// (composite.hip).  Floating-point contraction is switched off inside the function so that both
// translation units compile it to the same instruction sequence (bit-identical seeds).
#pragma once
#include "ngp_common.h"

// per-ray loss terms + backward seeds (losses.py:47-60, train.py:173, bg blend rendering.py:153-161)
__device__ __forceinline__ void nerf_loss_ray(float o, const float (&c)[3], const float (&g)[3], const float* __restrict__ bg,
                                              float lambda_o, float grad_scale, float inv_r, float inv_3r,
                                              float (&d_rgb)[3], float& d_o, float& l, float& se) {
#pragma clang fp contract(off)
    float go = 0.f, se_ray = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float b = bg ? bg[k] : 0.f;
        const float diff = c[k] + b * (1.0f - o) - g[k];
        se_ray += diff * diff;
        const float gr = 2.0f * diff * inv_3r;
        d_rgb[k] = gr * grad_scale;
        go -= gr * b;
    }
    const float oe = o + 1e-10f;
    const float lg = __logf(oe);
    l += se_ray * inv_3r + lambda_o * (-oe * lg) * inv_r;
    se += se_ray;
    go += lambda_o * (-(lg + 1.0f)) * inv_r;
    d_o = go * grad_scale;
}
