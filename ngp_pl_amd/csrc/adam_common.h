// The Adam update of ONE parameter, shared by every kernel that applies it to the hash table (optim.hip: the streaming kernels;
// hashgrid_bwd_binned.hip: the slice owners' write-out), so that they agree bit for bit: one expression tree, one set of compiler
// decisions.  Semantics: apex FusedAdam as the reference configures it (/root/reference/train.py:131-137; adam_w_mode = True,
// apex's default: DECOUPLED weight decay, p -= lr * wd * p next to the Adam term; optim.py refuses adam_w_mode=False), bias corrections bc1 = 1 - beta1^t, bc2 = 1 - beta2^t.
#pragma once

struct AdamCoef { float lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale; };

// g_raw: the gradient as stored (loss-scaled); inv_scale undoes the scale
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g_raw, const AdamCoef& c) {
    const float gk = g_raw * c.inv_scale;
    m = c.beta1 * m + (1.f - c.beta1) * gk;
    v = c.beta2 * v + (1.f - c.beta2) * gk * gk;
    const float denom = sqrtf(v / c.bc2) + c.eps;
    p = p - c.lr * ((m / c.bc1) / denom + c.wd * p);
}
