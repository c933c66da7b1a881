// Optimizer / AMP plumbing / loss seeds for gfx950: dense, HBM-bound streaming kernels.
//
//  * ngp_adam_step    apex FusedAdam semantics (train.py:131: lr 1e-2, eps 1e-15, adam_w_mode,
//                     bias correction) fused with gradient unscale, the f32->f16 parameter cast
//                     tiny-cuda-nn performs each forward, and gradient zeroing: ONE pass over
//                     the 11.4 M parameters (30 B/param with f16 grads) instead of four.
//  * ngp_nerf_loss    NeRFLoss (losses.py:47-60) + mean reduction (train.py:173) + background
//                     blend (rendering.py:153-161) with analytic backward seeds.
#include "ngp_common.h"
#include "adam_common.h"
#include "loss_common.h"
#include <hip/hip_fp16.h>

namespace {

typedef _Float16 h1;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Dynamic loss scale on the device (torch.cuda.amp.GradScaler's rule, which Lightning's precision=16 puts on top of tiny-cuda-nn's
// fixed 128 in the reference, train.py:274): state = {f32 scale[2], i32 growth_tracker[2]}.  The launches of a step read slot
// `slot` (the field backward multiplies its seeds by scale[slot], this kernel divides the gradients by it -- powers of two: exact);
// the first MLP workgroup writes slot ^ 1: scale * backoff and tracker 0 when the step's flag is raised, else tracker + 1, and
// scale * growth every `interval` clean steps.  Readers and the writer never share a word inside a launch.
struct LossScaler { float* state; int slot; float growth, backoff; int interval; float lo, hi; };

struct AdamHyper { float lr, beta1, beta2, eps, wd, bc1, bc2, inv_scale; const int32_t* found_inf; int zero_grad;
                   const int32_t* found_inf_dense;        // skip flag of the dense (grid) block: found_inf unless a caller gives it its own
                   int32_t* step_state; int slot;         // device-side counts of APPLIED steps (below), or NULL: bc1 / bc2 as given
                   LossScaler scaler; };                  // state == NULL: none

// Bias correction under a skip flag.  apex / GradScaler leave the optimizer's step count unchanged when a step is skipped; the host
// cannot know whether the device-side flag was raised without a sync, so the count of APPLIED steps lives next to the flag:
// step_state = 4 x i32 on the device, {MLP blocks: slot 0, slot 1, grid block: slot 0, slot 1} (the two flags of the sharded
// exchange decide their blocks separately, so each has its own count).  Launch number c (1-based) reads slot (c - 1) & 1 and its
// first workgroup of each kind writes slot c & 1 = applied + (skipped ? 0 : 1): readers and the writer never share an address
// inside a launch, launches are ordered by the stream.
__device__ __forceinline__ void adam_bias_from_state(AdamHyper& h, bool dense, bool writer) {
    if (h.scaler.state != nullptr) {
        const LossScaler& q = h.scaler;
        const float sc = q.state[q.slot];
        h.inv_scale = h.inv_scale / sc;
        if (writer && !dense && threadIdx.x == 0) {       // decided by the MLP blocks' flag (under a data-parallel exchange: the one every rank agrees on)
            int32_t* tr = reinterpret_cast<int32_t*>(q.state + 2);
            const bool skip = h.found_inf != nullptr && *h.found_inf != 0;
            float next = sc;
            int32_t t = tr[q.slot];
            if (skip) { next = fmaxf(sc * q.backoff, q.lo); t = 0; }
            else if (++t >= q.interval) { next = fminf(sc * q.growth, q.hi); t = 0; }
            q.state[q.slot ^ 1] = next; tr[q.slot ^ 1] = t;
        }
    }
    if (h.step_state == nullptr) return;
    int32_t* st = h.step_state + (dense ? 2 : 0);
    const int32_t applied = st[h.slot];
    const int32_t* flag = dense ? h.found_inf_dense : h.found_inf;
    const bool skip = flag != nullptr && *flag != 0;
    const float t = (float)(applied + 1);
    h.bc1 = 1.0f - powf(h.beta1, t); h.bc2 = 1.0f - powf(h.beta2, t);
    if (writer && threadIdx.x == 0) st[h.slot ^ 1] = skip ? applied : applied + 1;
}

// f32 sum -> the f16 the gradient table holds, SATURATED: a sum over thousands of samples can leave the f16 range when no single
// sample does (the overflow guard watches the samples); +-65504 instead of inf keeps Adam's moments finite.
__device__ __forceinline__ h1 sat_f16(float v) { return (h1)fminf(fmaxf(v, -65504.0f), 65504.0f); }

// The coarse dense levels of the grid, whose gradient the binned table backward leaves as K partial f32 tables per level
// (ngp_grid_partials, include/ngp_hip.h): device-side copy of the record.
struct GridPartials { long long value_end; uint32_t offset[8]; int k_split[8]; long long part_off[8]; const float2* partial; int n_levels; };

// Dense (streaming) update of n parameters by workgroups `block` of `n_blocks`, 4 parameters per thread and trip.
// MERGE: gradient values below gp.value_end are formed here from the K partial tables -- summed in part order in f32 and rounded to
// f16 exactly as merge_kernel (hashgrid_bwd_binned.hip) does, so the update is bit-identical to merge + this kernel without MERGE.
template <bool GRAD_F32, bool MERGE = false>
__device__ __forceinline__ void adam_dense(float* __restrict__ param, h1* __restrict__ param_h, void* __restrict__ grad,
                                           float* __restrict__ m, float* __restrict__ v, long long n4, long long n,
                                           const AdamHyper& hp, int block, int n_blocks, const GridPartials* gp = nullptr) {
    const AdamCoef coef = {hp.lr, hp.beta1, hp.beta2, hp.eps, hp.wd, hp.bc1, hp.bc2, hp.inv_scale};
    const bool skip = hp.found_inf_dense != nullptr && *hp.found_inf_dense != 0;
    const long long stride = (long long)n_blocks * blockDim.x;
    for (long long q = (long long)block * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const long long base = q * 4;
        float g[4];
        const int cnt = (int)((n - base) < 4 ? (n - base) : 4);
        if (cnt == 4) {
            if (MERGE && base < gp->value_end) {
                // two entries of one level (levels are multiples of 8 entries long): sum their K partials
                const uint32_t e0 = (uint32_t)(base >> 1);
                int l = 0;
                while (l + 1 < gp->n_levels && e0 >= gp->offset[l + 1]) ++l;
                const uint32_t size = gp->offset[l + 1] - gp->offset[l];
                const float2* __restrict__ p = gp->partial + gp->part_off[l] + (e0 - gp->offset[l]);
                float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
                for (int k = 0; k < gp->k_split[l]; ++k) {
                    const float4 t = *reinterpret_cast<const float4*>(p + (size_t)k * size);
                    a0 += t.x; b0 += t.y; a1 += t.z; b1 += t.w;
                }
                g[0] = (float)sat_f16(a0); g[1] = (float)sat_f16(b0); g[2] = (float)sat_f16(a1); g[3] = (float)sat_f16(b1);     // (as merge_kernel)
            } else if (GRAD_F32) {
                float4* gp4 = reinterpret_cast<float4*>(reinterpret_cast<float*>(grad) + base);
                const float4 t = *gp4; g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
                if (hp.zero_grad) *gp4 = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                half4_t* gp4 = reinterpret_cast<half4_t*>(reinterpret_cast<h1*>(grad) + base);
                const half4_t t = *gp4;
                // (an f16 gradient is finite unless a path that sums in f16 overflowed -- the one-pass fallback of oversized batches, a
                // reduce-scatter of the ranks' tables: +-65504 instead of inf / NaN keeps the moments finite; free on this stream)
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = fminf(fmaxf((float)t[k], -65504.0f), 65504.0f);
                if (hp.zero_grad) { const half4_t z = {0, 0, 0, 0}; *gp4 = z; }
            }
            if (skip) continue;
            float4 p = *reinterpret_cast<float4*>(param + base);
            float4 mm = *reinterpret_cast<float4*>(m + base);
            float4 vv = *reinterpret_cast<float4*>(v + base);
            float* pp = &p.x; float* mp = &mm.x; float* vp = &vv.x;
            half4_t ph;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                adam_one(pp[k], mp[k], vp[k], g[k], coef);
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
                if (GRAD_F32) { float* gp = reinterpret_cast<float*>(grad) + i; gk = *gp; if (hp.zero_grad) *gp = 0.f; }
                else { h1* gp = reinterpret_cast<h1*>(grad) + i; gk = fminf(fmaxf((float)*gp, -65504.0f), 65504.0f); if (hp.zero_grad) *gp = (h1)0; }
                if (skip) continue;
                float pk = param[i], mk = m[i], vk = v[i];
                adam_one(pk, mk, vk, gk, coef);
                m[i] = mk; v[i] = vk; param[i] = pk;
                if (param_h) param_h[i] = (h1)pk;
            }
        }
    }
}

template <bool GRAD_F32>
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ param, h1* __restrict__ param_h, void* __restrict__ grad,
            float* __restrict__ m, float* __restrict__ v, long long n4, long long n, AdamHyper hp) {
    adam_dense<GRAD_F32>(param, param_h, grad, m, v, n4, n, hp, blockIdx.x, gridDim.x);
}

// out[i] = sum_p partials[p][i].  32 columns x 8 row lanes per workgroup, 4 independent loads in
// flight per thread (a thread per column walking all rows serially measured 62 us for 256 x 10240
// on MI355X, 64 columns x 4 row lanes 21 us).
__global__ void __launch_bounds__(256)
reduce_partials_kernel(const float* __restrict__ partials, int n_partials, int n, float* __restrict__ out) {
    __shared__ float s_acc[8][32];
    const int c = threadIdx.x & 31, lane_row = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < n) {
        int p = lane_row;
        for (; p + 24 < n_partials; p += 32) {
            a0 += partials[(size_t)p * n + col]; a1 += partials[(size_t)(p + 8) * n + col];
            a2 += partials[(size_t)(p + 16) * n + col]; a3 += partials[(size_t)(p + 24) * n + col];
        }
        for (; p < n_partials; p += 8) a0 += partials[(size_t)p * n + col];
    }
    s_acc[lane_row][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (lane_row == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_acc[r][c];
        out[col] = t;
    }
}

// Adam for the small MLP blocks straight from the per-workgroup partial sums: one thread per
// parameter sums its column of `partials` (n_partials x n) and applies the update.
__device__ __forceinline__ void adam_from_partials(float* __restrict__ param, h1* __restrict__ param_h, const float* __restrict__ partials,
                                                   int n_partials, float* __restrict__ m, float* __restrict__ v, int n,
                                                   const AdamHyper& hp, int block, float (*s_acc)[32]) {
    const int c = threadIdx.x & 31, lane_row = threadIdx.x >> 5;
    const int i = block * 32 + c;
    float a0 = 0.f, a1 = 0.f;
    if (i < n) {
        int p = lane_row;
        for (; p + 8 < n_partials; p += 16) { a0 += partials[(size_t)p * n + i]; a1 += partials[(size_t)(p + 8) * n + i]; }
        for (; p < n_partials; p += 8) a0 += partials[(size_t)p * n + i];
    }
    s_acc[lane_row][c] = a0 + a1;
    __syncthreads();
    if (lane_row != 0 || i >= n) return;
    if (hp.found_inf != nullptr && *hp.found_inf != 0) return;
    float g = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) g += s_acc[r][c];
    g *= hp.inv_scale;
    const float mk = hp.beta1 * m[i] + (1.f - hp.beta1) * g;
    const float vk = hp.beta2 * v[i] + (1.f - hp.beta2) * g * g;
    const float pk = param[i] - hp.lr * ((mk / hp.bc1) / (sqrtf(vk / hp.bc2) + hp.eps) + hp.wd * param[i]);
    m[i] = mk; v[i] = vk; param[i] = pk;
    if (param_h) param_h[i] = (h1)pk;
}

__global__ void __launch_bounds__(256)
adam_partials_kernel(float* __restrict__ param, h1* __restrict__ param_h, const float* __restrict__ partials, int n_partials,
                     float* __restrict__ m, float* __restrict__ v, int n, AdamHyper hp) {
    __shared__ float s_acc[8][32];
    adam_from_partials(param, param_h, partials, n_partials, m, v, n, hp, blockIdx.x, s_acc);
}

// The whole field in ONE launch: the two MLP blocks (from their partial sums; latency-bound, a few hundred
// workgroups) are dispatched first and run underneath the HBM-bound stream over the grid parameters.
struct AdamMlp { float* param; h1* param_h; const float* partials; float* m; float* v; int n; int blocks; };
__global__ void __launch_bounds__(256)
adam_field_kernel(float* __restrict__ param, h1* __restrict__ param_h, void* __restrict__ grad16, float* __restrict__ m,
                  float* __restrict__ v, long long n4, long long n, AdamMlp a, AdamMlp b, int n_partials, AdamHyper hp) {
    __shared__ float s_acc[8][32];
    const int blk = blockIdx.x;
    const bool dense = blk >= a.blocks + b.blocks;
    adam_bias_from_state(hp, dense, blk == 0 || blk == a.blocks + b.blocks);
    if (blk < a.blocks) adam_from_partials(a.param, a.param_h, a.partials, n_partials, a.m, a.v, a.n, hp, blk, s_acc);
    else if (!dense) adam_from_partials(b.param, b.param_h, b.partials, n_partials, b.m, b.v, b.n, hp, blk - a.blocks, s_acc);
    else adam_dense<false>(param, param_h, grad16, m, v, n4, n, hp, blk - a.blocks - b.blocks, (int)gridDim.x - a.blocks - b.blocks);
}

// The same launch for a data-parallel rank that owns one PIECE of every chunk of the table (csrc/comm.hip, stepper tail): the
// table is exchanged in n_chunks chunks of world x piece values; this rank updates values [c * chunk + rank * piece, + piece) of
// every chunk c from the reduce-scatter's output, which holds the rank's pieces back to back.  piece is a multiple of 8 and
// n_grid of 16, so neither a piece boundary nor the end of the table falls inside a group of 4.
struct AdamPieces { long long piece, chunk, rank_off, n_grid; int n_chunks; };
__device__ __forceinline__ void adam_dense_pieces(float* __restrict__ param, h1* __restrict__ param_h, const h1* __restrict__ grad,
                                                  float* __restrict__ m, float* __restrict__ v, const AdamPieces pc,
                                                  const AdamHyper& hp, int block, int n_blocks) {
    const AdamCoef coef = {hp.lr, hp.beta1, hp.beta2, hp.eps, hp.wd, hp.bc1, hp.bc2, hp.inv_scale};
    if (hp.found_inf_dense != nullptr && *hp.found_inf_dense != 0) return;
    const long long n4 = (long long)pc.n_chunks * pc.piece / 4, stride = (long long)n_blocks * blockDim.x;
    for (long long q = (long long)block * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const long long i = q * 4, c = i / pc.piece, base = c * pc.chunk + pc.rank_off + (i - c * pc.piece);
        if (base >= pc.n_grid) continue;                                   // padding behind the table
        const half4_t t = *reinterpret_cast<const half4_t*>(grad + i);
        float4 p = *reinterpret_cast<float4*>(param + base);
        float4 mm = *reinterpret_cast<float4*>(m + base);
        float4 vv = *reinterpret_cast<float4*>(v + base);
        float* pp = &p.x; float* mp = &mm.x; float* vp = &vv.x;
        half4_t ph;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            adam_one(pp[k], mp[k], vp[k], (float)t[k], coef);
            ph[k] = (h1)pp[k];
        }
        *reinterpret_cast<float4*>(param + base) = p;
        *reinterpret_cast<float4*>(m + base) = mm;
        *reinterpret_cast<float4*>(v + base) = vv;
        *reinterpret_cast<half4_t*>(param_h + base) = ph;
    }
}

__global__ void __launch_bounds__(256)
adam_field_pieces_kernel(float* __restrict__ param, h1* __restrict__ param_h, const h1* __restrict__ shard16, float* __restrict__ m,
                         float* __restrict__ v, AdamPieces pc, AdamMlp a, AdamMlp b, int n_partials, AdamHyper hp) {
    __shared__ float s_acc[8][32];
    const int blk = blockIdx.x;
    const bool dense = blk >= a.blocks + b.blocks;
    adam_bias_from_state(hp, dense, blk == 0 || blk == a.blocks + b.blocks);
    if (blk < a.blocks) adam_from_partials(a.param, a.param_h, a.partials, n_partials, a.m, a.v, a.n, hp, blk, s_acc);
    else if (!dense) adam_from_partials(b.param, b.param_h, b.partials, n_partials, b.m, b.v, b.n, hp, blk - a.blocks, s_acc);
    else adam_dense_pieces(param, param_h, shard16, m, v, pc, hp, blk - a.blocks - b.blocks, (int)gridDim.x - a.blocks - b.blocks);
}

__global__ void __launch_bounds__(256)
adam_field_merge_kernel(float* __restrict__ param, h1* __restrict__ param_h, void* __restrict__ grad16, float* __restrict__ m,
                        float* __restrict__ v, long long n4, long long n, AdamMlp a, AdamMlp b, int n_partials, AdamHyper hp, GridPartials gp) {
    __shared__ float s_acc[8][32];
    const int blk = blockIdx.x;
    const bool dense = blk >= a.blocks + b.blocks;
    adam_bias_from_state(hp, dense, blk == 0 || blk == a.blocks + b.blocks);
    if (blk < a.blocks) adam_from_partials(a.param, a.param_h, a.partials, n_partials, a.m, a.v, a.n, hp, blk, s_acc);
    else if (!dense) adam_from_partials(b.param, b.param_h, b.partials, n_partials, b.m, b.v, b.n, hp, blk - a.blocks, s_acc);
    else adam_dense<false, true>(param, param_h, grad16, m, v, n4, n, hp, blk - a.blocks - b.blocks, (int)gridDim.x - a.blocks - b.blocks, &gp);
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

// OVERWRITE mode (n_rays <= 16384): workgroups park their partial sums in a library-owned scratch,
// the last one to finish (self-resetting ticket: atomicInc wraps at the workgroup count) adds them
// in workgroup order and WRITES loss / sq_err -- no zero-fill by the caller, no float atomics,
// deterministic.  The scratch is shared by all launches of this kernel: callers must not run two
// of them concurrently on different streams.
constexpr int LOSS_MAX_BLOCKS = 64;
__device__ unsigned int g_loss_ticket = 0;
__device__ float g_loss_partial[2 * LOSS_MAX_BLOCKS];

template <bool OVERWRITE>
__global__ void __launch_bounds__(256)
nerf_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ opacity, const float* __restrict__ gt,
                 const float* __restrict__ bg, float lambda_o, float grad_scale, int n_rays,
                 float* __restrict__ loss, float* __restrict__ sq_err,
                 float* __restrict__ dL_drgb, float* __restrict__ dL_dopacity) {
    float l = 0.f, se = 0.f;
    const float inv_r = 1.0f / (float)n_rays, inv_3r = 1.0f / (3.0f * (float)n_rays);
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rays) {
        const float c[3] = {rgb[3 * r], rgb[3 * r + 1], rgb[3 * r + 2]}, g[3] = {gt[3 * r], gt[3 * r + 1], gt[3 * r + 2]};
        float d_rgb[3], d_o;
        nerf_loss_ray(opacity[r], c, g, bg, lambda_o, grad_scale, inv_r, inv_3r, d_rgb, d_o, l, se);
        dL_drgb[3 * r] = d_rgb[0]; dL_drgb[3 * r + 1] = d_rgb[1]; dL_drgb[3 * r + 2] = d_rgb[2];
        dL_dopacity[r] = d_o;
    }
    l = ngp_wave_sum(l); se = ngp_wave_sum(se);
    __shared__ float s_l[4], s_e[4];
    __shared__ bool s_last;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_l[w] = l; s_e[w] = se; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tl = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]), te = (s_e[0] + s_e[1]) + (s_e[2] + s_e[3]);
        if (OVERWRITE) {
            g_loss_partial[2 * blockIdx.x] = tl; g_loss_partial[2 * blockIdx.x + 1] = te;
            __threadfence();
            s_last = atomicInc(&g_loss_ticket, gridDim.x - 1) == gridDim.x - 1;
        } else {
            atomicAdd(loss, tl); if (sq_err) atomicAdd(sq_err, te);
        }
    }
    if (!OVERWRITE) return;
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        __threadfence();
        float a = 0.f, b = 0.f;
        if (threadIdx.x < gridDim.x) {
            a = __builtin_nontemporal_load(&g_loss_partial[2 * threadIdx.x]);
            b = __builtin_nontemporal_load(&g_loss_partial[2 * threadIdx.x + 1]);
        }
        a = ngp_wave_sum(a); b = ngp_wave_sum(b);
        if (threadIdx.x == 0) { *loss = a; if (sq_err) *sq_err = b; }
    }
}


// NeRFLoss's per-element terms (losses.py:47-60: the module returns a dictionary of UNREDUCED terms, train.py:173 takes their
// means) and their backward, one launch each instead of ~6 + ~10 elementwise torch kernels.
__global__ void __launch_bounds__(256)
nerf_loss_terms_fw_kernel(const float* __restrict__ rgb, const float* __restrict__ opacity, const float* __restrict__ gt,
                          float lambda_o, int n_rays, float* __restrict__ sq, float* __restrict__ ent) {
#pragma clang fp contract(off)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float d = rgb[3 * r + k] - gt[3 * r + k]; sq[3 * r + k] = d * d; }
    const float o = opacity[r] + 1e-10f;
    ent[r] = lambda_o * (-o * logf(o));
}
__global__ void __launch_bounds__(256)
nerf_loss_terms_bw_kernel(const float* __restrict__ g_sq, int g_sq_bcast, const float* __restrict__ g_ent, int g_ent_bcast,
                          const float* __restrict__ rgb, const float* __restrict__ opacity, const float* __restrict__ gt, float lambda_o,
                          int n_rays, float* __restrict__ g_rgb, float* __restrict__ g_opacity) {
#pragma clang fp contract(off)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) g_rgb[3 * r + k] = g_sq[g_sq_bcast ? 0 : 3 * r + k] * (2.0f * (rgb[3 * r + k] - gt[3 * r + k]));
    const float o = opacity[r] + 1e-10f;
    g_opacity[r] = g_ent[g_ent_bcast ? 0 : r] * (lambda_o * (-(logf(o) + 1.0f)));
}

// render()'s background blend (rendering.py:153-161, rgb + bg (1 - opacity)) and its backward onto the opacity seed, one launch each
__global__ void __launch_bounds__(256)
bg_blend_kernel(const float* __restrict__ rgb, const float* __restrict__ opacity, const float* __restrict__ bg, int n_rays,
                float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float tr = 1.0f - opacity[r];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[3 * r + k] = fmaf(bg[k], tr, rgb[3 * r + k]);
}
__global__ void __launch_bounds__(256)
bg_blend_bw_kernel(const float* __restrict__ g_rgb, const float* __restrict__ g_opacity, const float* __restrict__ bg, int n_rays,
                   float* __restrict__ g_opacity_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float g = g_opacity ? g_opacity[r] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) g = fmaf(-g_rgb[3 * r + k], bg[k], g);      // (explicit: composite_train_bw_kernel<true> folds the same line in)
    g_opacity_out[r] = g;
}

// GPU-resident batch sampler: the reference draws img/pix indices with np.random.choice in 16
// dataloader workers, gathers rays[img, pix] from a CPU tensor and ships the batch over PCIe
// (datasets/base.py:22-35, train.py:141-146), then forms rays on the GPU (train.py:78-91,
// ray_utils.py:46-70).  Here one kernel draws the indices (counter-based hash RNG), gathers the
// ground-truth colour and rotates the pixel direction by the camera pose.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) { return ngp_pcg_hash(v); }
__global__ void __launch_bounds__(256)
sample_rays_kernel(const float* __restrict__ poses, const float* __restrict__ directions, const float* __restrict__ images,
                   int n_images, int n_pixels, int n, uint32_t seed_lo, uint32_t seed_hi,
                   float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ rgb,
                   float* __restrict__ noise, int32_t* __restrict__ img_idx, int32_t* __restrict__ pix_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t base = ngp_rng_key(seed_lo, seed_hi, (uint32_t)i);
    const uint32_t r0 = pcg_hash(base), r1 = pcg_hash(base + 1u), r2 = pcg_hash(base + 2u);
    const int img = (int)(((uint64_t)r0 * (uint64_t)n_images) >> 32);     // uniform in [0, n_images)
    const int pix = (int)(((uint64_t)r1 * (uint64_t)n_pixels) >> 32);
    const float* P = poses + 12 * (size_t)img;                            // (3,4) row-major c2w
    const float dx = directions[3 * (size_t)pix], dy = directions[3 * (size_t)pix + 1], dz = directions[3 * (size_t)pix + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_d[3 * (size_t)i + k] = (dx * P[4 * k] + dy * P[4 * k + 1]) + dz * P[4 * k + 2];   // sum order of (d * R).sum(-1)
        rays_o[3 * (size_t)i + k] = P[4 * k + 3];
        rgb[3 * (size_t)i + k] = images[((size_t)img * n_pixels + pix) * 3 + k];
    }
    if (noise) noise[i] = (float)(r2 >> 8) * (1.0f / 16777216.0f);         // [0,1), 24 bits like torch.rand
    if (img_idx) { img_idx[i] = img; pix_idx[i] = pix; }
}


// datasets/ray_utils.py:50-74 get_rays for ONE camera: rays_d = directions @ c2w[:, :3].T (not normalised), rays_o = camera centre
// repeated -- one launch instead of torch's GEMM + expand + two contiguous copies (60 us of a 1.6 ms frame in the FPS protocol,
// which times ray generation together with render(), test.ipynb cell 2).
__global__ void __launch_bounds__(256)
get_rays_kernel(const float* __restrict__ directions, const float* __restrict__ c2w, int n,
                float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = directions[3 * (size_t)i], dy = directions[3 * (size_t)i + 1], dz = directions[3 * (size_t)i + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_d[3 * (size_t)i + k] = (dx * c2w[4 * k] + dy * c2w[4 * k + 1]) + dz * c2w[4 * k + 2];
        rays_o[3 * (size_t)i + k] = c2w[4 * k + 3];
    }
}

// GradScaler's inf check (torch.amp.GradScaler.unscale_ -> _amp_foreach_non_finite_check_and_unscale_) on a native
// gradient buffer: flag[0] |= 1 if any element is inf/NaN.  16 bytes per lane and trip; f16 exponent all-ones test on the raw bits.
__global__ void __launch_bounds__(256)
found_inf_kernel(const u32x4* __restrict__ g, long long n16, const unsigned short* __restrict__ tail, int n_tail, int is_half,
                 int32_t* __restrict__ flag) {
    bool bad = false;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += stride) {
        const u32x4 w = __builtin_nontemporal_load(g + q);
        const uint32_t x[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (is_half) bad |= ((x[k] & 0x7c00u) == 0x7c00u) | ((x[k] & 0x7c000000u) == 0x7c000000u);
            else bad |= (x[k] & 0x7f800000u) == 0x7f800000u;
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) {          // the < 16 trailing bytes, as 16-bit words
        if (is_half) bad |= (tail[threadIdx.x] & 0x7c00u) == 0x7c00u;
        else if (threadIdx.x & 1) bad |= (tail[threadIdx.x] & 0x7f80u) == 0x7f80u;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// The same check over TWO buffers in one launch (the reduced grid gradient and the reduced MLP sums), with the flag of the
// NEXT step cleared on the side: callers alternate between two flags, so no memset launch is needed in steady state.
__global__ void __launch_bounds__(256)
found_inf2_kernel(const u32x4* __restrict__ a, long long a16, int a_half, const u32x4* __restrict__ b, long long b16, int b_half,
                  int32_t* __restrict__ flag, int32_t* __restrict__ flag_clear) {
    bool bad = false;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < a16 + b16; q += stride) {
        const bool in_a = q < a16;
        const u32x4 w = in_a ? __builtin_nontemporal_load(a + q) : __builtin_nontemporal_load(b + (q - a16));
        const int is_half = in_a ? a_half : b_half;
        const uint32_t x[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (is_half) bad |= ((x[k] & 0x7c00u) == 0x7c00u) | ((x[k] & 0x7c000000u) == 0x7c000000u);
            else bad |= (x[k] & 0x7f800000u) == 0x7f800000u;
        }
    }
    if (flag_clear && blockIdx.x == 0 && threadIdx.x == 0) *flag_clear = 0;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// out[0:n_a] = column sums of partials_a (n_partials, n_a), out[n_a:n_a+n_b] = column sums of partials_b (n_partials, n_b): both MLP
// blocks' per-workgroup partial sums in one launch (reduce_partials_kernel's scheme).
__global__ void __launch_bounds__(256)
reduce_partials2_kernel(const float* __restrict__ pa, int n_a, const float* __restrict__ pb, int n_b, int n_partials, float* __restrict__ out) {
    __shared__ float s_acc[8][32];
    const int c = threadIdx.x & 31, lane_row = threadIdx.x >> 5;
    const int blocks_a = (n_a + 31) / 32;
    const bool first = (int)blockIdx.x < blocks_a;
    const float* __restrict__ partials = first ? pa : pb;
    const int n = first ? n_a : n_b;
    const int col = (first ? blockIdx.x : blockIdx.x - blocks_a) * 32 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < n) {
        int p = lane_row;
        for (; p + 24 < n_partials; p += 32) {
            a0 += partials[(size_t)p * n + col]; a1 += partials[(size_t)(p + 8) * n + col];
            a2 += partials[(size_t)(p + 16) * n + col]; a3 += partials[(size_t)(p + 24) * n + col];
        }
        for (; p < n_partials; p += 8) a0 += partials[(size_t)p * n + col];
    }
    s_acc[lane_row][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (lane_row == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_acc[r][c];
        out[(first ? 0 : n_a) + col] = t;
    }
}

// (csrc/ngp_internal.h: ngp_adam_use_loss_scaler) the scaler of the NEXT optimizer launch this thread enqueues; consumed by it
thread_local LossScaler t_next_scaler = {nullptr, 0, 2.0f, 0.5f, 2000, 1.0f, 1.0f};

AdamHyper adam_hyper(float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale, const int32_t* found_inf) {
    AdamHyper hp;
    hp.scaler = t_next_scaler;
    t_next_scaler.state = nullptr;
    hp.lr = lr; hp.beta1 = beta1; hp.beta2 = beta2; hp.eps = eps; hp.wd = wd;
    hp.bc1 = 1.0f - powf(beta1, (float)step); hp.bc2 = 1.0f - powf(beta2, (float)step);
    hp.inv_scale = 1.0f / grad_scale; hp.found_inf = found_inf; hp.found_inf_dense = found_inf; hp.zero_grad = 1;
    hp.step_state = nullptr; hp.slot = 0;
    return hp;
}


// out = sum over the world slices, rank order, f32, one rounding to f16 (the direct exchange's "reduce" half).  8 values per thread.
__global__ void __launch_bounds__(256)
sum_slices_kernel(const uint4* __restrict__ own, const uint4* __restrict__ stage, int world, int rank, long long n8, uint4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < world; ++q) {
        const uint4 v = (q == rank) ? own[i] : stage[(long long)q * n8 + i];
        const _Float16* h = reinterpret_cast<const _Float16*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)h[k];
    }
    uint4 o;
    _Float16* oh = reinterpret_cast<_Float16*>(&o);
#pragma unroll
    for (int k = 0; k < 8; ++k) oh[k] = (_Float16)acc[k];
    out[i] = o;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

// (csrc/ngp_internal.h) Hands the NEXT optimizer launch enqueued by this thread (any ngp_adam_step_field* entry point) a device-side
// dynamic loss scale: state = {f32 scale[2], i32 growth_tracker[2]} on the device, slot = the half this step's launches read.
int ngp_adam_use_loss_scaler(float* state, int slot, float growth_factor, float backoff_factor, int growth_interval, float min_scale, float max_scale) {
    if (state == nullptr) { t_next_scaler.state = nullptr; return 0; }
    if ((slot & ~1) || !(growth_factor >= 1.0f) || !(backoff_factor > 0.0f && backoff_factor <= 1.0f) || growth_interval < 1 || !(min_scale > 0.f) || !(max_scale >= min_scale))
        return NGP_EINVAL;
    t_next_scaler = {state, slot, growth_factor, backoff_factor, growth_interval, min_scale, max_scale};
    return 0;
}

int ngp_adam_step(float* param, ngp_half* param_h, void* grad, int grad_is_f32, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  const int32_t* found_inf, ngp_stream_t stream) {
    if (n < 0 || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(param); NGP_CHECK_PTR(grad); NGP_CHECK_PTR(m); NGP_CHECK_PTR(v);
    const AdamHyper hp = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf);
    const long long n4 = (n + 3) / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    if (grad_is_f32)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, ngp_stream(stream), param, (h1*)param_h, grad, m, v,
                           n4, (long long)n, hp);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, ngp_stream(stream), param, (h1*)param_h, grad, m, v,
                           n4, (long long)n, hp);
    return NGP_LAUNCH_RESULT();
}

int ngp_adam_step_partials(float* param, ngp_half* param_h, const float* partials, int n_partials, float* m, float* v,
                           int n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           float grad_scale, const int32_t* found_inf, ngp_stream_t stream) {
    if (n < 0 || n_partials < 0 || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(param); NGP_CHECK_PTR(m); NGP_CHECK_PTR(v);
    if (n_partials > 0) NGP_CHECK_PTR(partials);
    const AdamHyper hp = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf);
    hipLaunchKernelGGL(adam_partials_kernel, dim3(ngp_div_up(n, 32)), dim3(256), 0, ngp_stream(stream), param, (h1*)param_h, partials,
                       n_partials, m, v, n, hp);
    return NGP_LAUNCH_RESULT();
}

static int adam_step_field_impl(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad, float* grid_m, float* grid_v, int64_t n_grid,
                                float* density_param, ngp_half* density_param_h, const float* density_partials, float* density_m,
                                float* density_v, int n_density, float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                                float* rgb_m, float* rgb_v, int n_rgb, int n_partials, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, float grad_scale, int zero_grid_grad, const int32_t* found_inf,
                                const int32_t* found_inf_grid, int32_t* step_state, ngp_stream_t stream) {
    // n_grid == 0: the MLP blocks only (a data-parallel rank whose shard of the table is empty)
    if (n_grid < 0 || n_density <= 0 || n_rgb <= 0 || n_partials < 0 || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    if (n_grid > 0) { NGP_CHECK_PTR(grid_param); NGP_CHECK_PTR(grid_grad); NGP_CHECK_PTR(grid_m); NGP_CHECK_PTR(grid_v); }
    NGP_CHECK_PTR(density_param); NGP_CHECK_PTR(density_m); NGP_CHECK_PTR(density_v);
    NGP_CHECK_PTR(rgb_param); NGP_CHECK_PTR(rgb_m); NGP_CHECK_PTR(rgb_v);
    if (n_partials > 0) { NGP_CHECK_PTR(density_partials); NGP_CHECK_PTR(rgb_partials); }
    AdamHyper hp = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf);
    hp.found_inf_dense = found_inf_grid;
    hp.zero_grad = zero_grid_grad != 0;
    hp.step_state = step_state; hp.slot = (step - 1) & 1;
    const long long n4 = (n_grid + 3) / 4;
    const int dense_blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    const AdamMlp a = {density_param, (h1*)density_param_h, density_partials, density_m, density_v, n_density, ngp_div_up(n_density, 32)};
    const AdamMlp b = {rgb_param, (h1*)rgb_param_h, rgb_partials, rgb_m, rgb_v, n_rgb, ngp_div_up(n_rgb, 32)};
    hipLaunchKernelGGL(adam_field_kernel, dim3(a.blocks + b.blocks + dense_blocks), dim3(256), 0, ngp_stream(stream), grid_param,
                       (h1*)grid_param_h, (void*)grid_grad, grid_m, grid_v, n4, (long long)n_grid, a, b, n_partials, hp);
    return NGP_LAUNCH_RESULT();
}

int ngp_adam_step_field(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad, float* grid_m, float* grid_v, int64_t n_grid,
                        float* density_param, ngp_half* density_param_h, const float* density_partials, float* density_m,
                        float* density_v, int n_density, float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                        float* rgb_m, float* rgb_v, int n_rgb, int n_partials, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int zero_grid_grad, const int32_t* found_inf, int32_t* step_state,
                        ngp_stream_t stream) {
    if (n_grid <= 0) return NGP_EINVAL;
    return adam_step_field_impl(grid_param, grid_param_h, grid_grad, grid_m, grid_v, n_grid, density_param, density_param_h, density_partials,
                                density_m, density_v, n_density, rgb_param, rgb_param_h, rgb_partials, rgb_m, rgb_v, n_rgb, n_partials, lr,
                                beta1, beta2, eps, weight_decay, step, grad_scale, zero_grid_grad, found_inf, found_inf, step_state, stream);
}

int ngp_adam_step_field_merge(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad, float* grid_m, float* grid_v, int64_t n_grid,
                              float* density_param, ngp_half* density_param_h, const float* density_partials, float* density_m,
                              float* density_v, int n_density, float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                              float* rgb_m, float* rgb_v, int n_rgb, int n_partials, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, const int32_t* found_inf, int32_t* step_state,
                              const ngp_grid_partials* partials, ngp_stream_t stream) {
    if (!partials) return NGP_EINVAL;
    const ngp_grid_partials& q = *partials;
    if (q.n_levels < 0 || q.n_levels > 8 || q.value_end < 0 || q.value_end > n_grid || (q.value_end & 3)) return NGP_EINVAL;
    if (q.n_levels == 0)
        return ngp_adam_step_field(grid_param, grid_param_h, grid_grad, grid_m, grid_v, n_grid, density_param, density_param_h, density_partials, density_m,
                                   density_v, n_density, rgb_param, rgb_param_h, rgb_partials, rgb_m, rgb_v, n_rgb, n_partials, lr, beta1, beta2, eps,
                                   weight_decay, step, grad_scale, 0, found_inf, step_state, stream);
    if (n_grid <= 0 || n_density <= 0 || n_rgb <= 0 || n_partials < 0 || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    NGP_CHECK_PTR(grid_param); NGP_CHECK_PTR(grid_grad); NGP_CHECK_PTR(grid_m); NGP_CHECK_PTR(grid_v); NGP_CHECK_PTR(q.partial);
    NGP_CHECK_PTR(density_param); NGP_CHECK_PTR(density_m); NGP_CHECK_PTR(density_v);
    NGP_CHECK_PTR(rgb_param); NGP_CHECK_PTR(rgb_m); NGP_CHECK_PTR(rgb_v);
    if (n_partials > 0) { NGP_CHECK_PTR(density_partials); NGP_CHECK_PTR(rgb_partials); }
    GridPartials gp;
    gp.value_end = q.value_end; gp.n_levels = q.n_levels; gp.partial = reinterpret_cast<const float2*>(q.partial);
    for (int l = 0; l < 8; ++l) {
        gp.offset[l] = q.offset[l < q.n_levels ? l : q.n_levels];
        gp.k_split[l] = l < q.n_levels ? q.k_split[l] : 1;
        gp.part_off[l] = l < q.n_levels ? q.part_off[l] : 0;
        if (l < q.n_levels && (((q.offset[l + 1] - q.offset[l]) & 1u) || (q.part_off[l] & 1) || q.k_split[l] < 1)) return NGP_EINVAL;   // 16-byte loads of entry pairs
    }
    if (q.n_levels == 8) return NGP_EUNSUP;                                   // offset[l + 1] of the last level would not fit the device record
    gp.offset[q.n_levels] = q.offset[q.n_levels];
    if (2 * (int64_t)q.offset[q.n_levels] != q.value_end) return NGP_EINVAL;
    AdamHyper hp = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf);
    hp.zero_grad = 0;
    hp.step_state = step_state; hp.slot = (step - 1) & 1;
    const long long n4 = (n_grid + 3) / 4;
    const int dense_blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    const AdamMlp a = {density_param, (h1*)density_param_h, density_partials, density_m, density_v, n_density, ngp_div_up(n_density, 32)};
    const AdamMlp b = {rgb_param, (h1*)rgb_param_h, rgb_partials, rgb_m, rgb_v, n_rgb, ngp_div_up(n_rgb, 32)};
    hipLaunchKernelGGL(adam_field_merge_kernel, dim3(a.blocks + b.blocks + dense_blocks), dim3(256), 0, ngp_stream(stream), grid_param,
                       (h1*)grid_param_h, (void*)grid_grad, grid_m, grid_v, n4, (long long)n_grid, a, b, n_partials, hp, gp);
    return NGP_LAUNCH_RESULT();
}

int ngp_adam_step_field_shard(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad, float* grid_m, float* grid_v, int64_t n_shard,
                              float* density_param, ngp_half* density_param_h, const float* density_partials, float* density_m,
                              float* density_v, int n_density, float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                              float* rgb_m, float* rgb_v, int n_rgb, int n_partials, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, const int32_t* found_inf_mlp,
                              const int32_t* found_inf_shard, int32_t* step_state, ngp_stream_t stream) {
    return adam_step_field_impl(grid_param, grid_param_h, grid_grad, grid_m, grid_v, n_shard, density_param, density_param_h, density_partials,
                                density_m, density_v, n_density, rgb_param, rgb_param_h, rgb_partials, rgb_m, rgb_v, n_rgb, n_partials, lr,
                                beta1, beta2, eps, weight_decay, step, grad_scale, 0, found_inf_mlp, found_inf_shard, step_state, stream);
}

int ngp_adam_step_field_pieces(float* grid_param, ngp_half* grid_param_h, const ngp_half* shard_grad, float* grid_m, float* grid_v,
                               int64_t n_grid, int64_t piece, int32_t n_chunks, int32_t world, int32_t rank,
                               float* density_param, ngp_half* density_param_h, const float* density_partials, float* density_m,
                               float* density_v, int n_density, float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                               float* rgb_m, float* rgb_v, int n_rgb, int n_partials, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float grad_scale, const int32_t* found_inf_mlp,
                               const int32_t* found_inf_shard, int32_t* step_state, ngp_stream_t stream) {
    if (n_grid <= 0 || (n_grid & 15) || piece < 8 || (piece & 7) || n_chunks < 1 || n_chunks > 8 || world < 1 || rank < 0 || rank >= world) return NGP_EINVAL;
    if ((long long)n_chunks * world * piece < n_grid) return NGP_EINVAL;                   // the chunks must cover the table
    if (n_density <= 0 || n_rgb <= 0 || n_partials < 0 || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    NGP_CHECK_PTR(grid_param); NGP_CHECK_PTR(grid_param_h); NGP_CHECK_PTR(shard_grad); NGP_CHECK_PTR(grid_m); NGP_CHECK_PTR(grid_v);
    NGP_CHECK_PTR(density_param); NGP_CHECK_PTR(density_m); NGP_CHECK_PTR(density_v);
    NGP_CHECK_PTR(rgb_param); NGP_CHECK_PTR(rgb_m); NGP_CHECK_PTR(rgb_v);
    if (n_partials > 0) { NGP_CHECK_PTR(density_partials); NGP_CHECK_PTR(rgb_partials); }
    AdamHyper hp = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf_mlp);
    hp.found_inf_dense = found_inf_shard;
    hp.zero_grad = 0;
    hp.step_state = step_state; hp.slot = (step - 1) & 1;
    const AdamPieces pc = {(long long)piece, (long long)world * piece, (long long)rank * piece, (long long)n_grid, n_chunks};
    const long long n4 = (long long)n_chunks * piece / 4;
    const int dense_blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    const AdamMlp a = {density_param, (h1*)density_param_h, density_partials, density_m, density_v, n_density, ngp_div_up(n_density, 32)};
    const AdamMlp b = {rgb_param, (h1*)rgb_param_h, rgb_partials, rgb_m, rgb_v, n_rgb, ngp_div_up(n_rgb, 32)};
    hipLaunchKernelGGL(adam_field_pieces_kernel, dim3(a.blocks + b.blocks + dense_blocks), dim3(256), 0, ngp_stream(stream), grid_param,
                       (h1*)grid_param_h, (const h1*)shard_grad, grid_m, grid_v, pc, a, b, n_partials, hp);
    return NGP_LAUNCH_RESULT();
}

int ngp_get_rays(const float* directions, const float* c2w, int n, float* rays_o, float* rays_d, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(directions); NGP_CHECK_PTR(c2w); NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d);
    hipLaunchKernelGGL(get_rays_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, ngp_stream(stream), directions, c2w, n, rays_o, rays_d);
    return NGP_LAUNCH_RESULT();
}

int ngp_found_inf(const void* grad, int grad_is_f32, int64_t n, int32_t* flag, int reset, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(flag);
    hipStream_t st = ngp_stream(stream);
    if (reset) { hipError_t e = hipMemsetAsync(flag, 0, sizeof(int32_t), st); if (e != hipSuccess) return (int)e; }
    if (n == 0) return 0;
    NGP_CHECK_PTR(grad);
    if (reinterpret_cast<uintptr_t>(grad) & 15) return NGP_EINVAL;
    const long long bytes = (long long)n * (grad_is_f32 ? 4 : 2), n16 = bytes / 16;
    const int n_tail = (int)((bytes - n16 * 16) / 2);
    const int blocks = (int)((n16 + 255) / 256 < 2048 ? ((n16 + 255) / 256 > 0 ? (n16 + 255) / 256 : 1) : 2048);
    hipLaunchKernelGGL(found_inf_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const u32x4*>(grad), n16,
                       reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(grad) + n16 * 16), n_tail, grad_is_f32 ? 0 : 1, flag);
    return NGP_LAUNCH_RESULT();
}

int ngp_found_inf2(const void* grad_a, int a_is_f32, int64_t n_a, const void* grad_b, int b_is_f32, int64_t n_b, int32_t* flag,
                   int32_t* flag_clear, ngp_stream_t stream) {
    if (n_a < 0 || n_b < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(flag);
    if (n_a > 0) NGP_CHECK_PTR(grad_a);
    if (n_b > 0) NGP_CHECK_PTR(grad_b);
    const long long bytes_a = (long long)n_a * (a_is_f32 ? 4 : 2), bytes_b = (long long)n_b * (b_is_f32 ? 4 : 2);
    if ((reinterpret_cast<uintptr_t>(grad_a) & 15) || (reinterpret_cast<uintptr_t>(grad_b) & 15) || (bytes_a & 15) || (bytes_b & 15)) return NGP_EINVAL;
    const long long n16 = (bytes_a + bytes_b) / 16;
    const int blocks = (int)((n16 + 255) / 256 < 2048 ? ((n16 + 255) / 256 > 0 ? (n16 + 255) / 256 : 1) : 2048);
    hipLaunchKernelGGL(found_inf2_kernel, dim3(blocks), dim3(256), 0, ngp_stream(stream), reinterpret_cast<const u32x4*>(grad_a), bytes_a / 16,
                       a_is_f32 ? 0 : 1, reinterpret_cast<const u32x4*>(grad_b), bytes_b / 16, b_is_f32 ? 0 : 1, flag, flag_clear);
    return NGP_LAUNCH_RESULT();
}

int ngp_reduce_partials2(const float* partials_a, int n_a, const float* partials_b, int n_b, int n_partials, float* out, ngp_stream_t stream) {
    if (n_partials < 0 || n_a < 0 || n_b < 0) return NGP_EINVAL;
    if (n_a + n_b == 0) return 0;
    NGP_CHECK_PTR(out);
    if (n_partials > 0) { NGP_CHECK_PTR(partials_a); NGP_CHECK_PTR(partials_b); }
    hipLaunchKernelGGL(reduce_partials2_kernel, dim3(ngp_div_up(n_a, 32) + ngp_div_up(n_b, 32)), dim3(256), 0, ngp_stream(stream),
                       partials_a, n_a, partials_b, n_b, n_partials, out);
    return NGP_LAUNCH_RESULT();
}

int ngp_reduce_partials(const float* partials, int n_partials, int n, float* out, ngp_stream_t stream) {
    if (n_partials < 0 || n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(out);
    if (n_partials > 0) NGP_CHECK_PTR(partials);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ngp_div_up(n, 32)), dim3(256), 0, ngp_stream(stream), partials, n_partials, n, out);
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
    if (n_rays <= 256 * LOSS_MAX_BLOCKS)   // overwrite mode: the caller does not have to zero loss / sq_err
        hipLaunchKernelGGL(nerf_loss_kernel<true>, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                           rgb, opacity, gt_rgb, bg, lambda_opacity, grad_scale, n_rays, loss, sq_err, dL_drgb, dL_dopacity);
    else {
        hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), ngp_stream(stream));
        if (e == hipSuccess && sq_err) e = hipMemsetAsync(sq_err, 0, sizeof(float), ngp_stream(stream));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nerf_loss_kernel<false>, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                           rgb, opacity, gt_rgb, bg, lambda_opacity, grad_scale, n_rays, loss, sq_err, dL_drgb, dL_dopacity);
    }
    return NGP_LAUNCH_RESULT();
}

int ngp_nerf_loss_terms_fw(const float* rgb, const float* opacity, const float* gt_rgb, float lambda_opacity, int n_rays,
                           float* sq_err, float* entropy, ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(gt_rgb); NGP_CHECK_PTR(sq_err); NGP_CHECK_PTR(entropy);
    hipLaunchKernelGGL(nerf_loss_terms_fw_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream), rgb, opacity, gt_rgb,
                       lambda_opacity, n_rays, sq_err, entropy);
    return NGP_LAUNCH_RESULT();
}

int ngp_nerf_loss_terms_bw(const float* g_sq_err, int g_sq_err_is_scalar, const float* g_entropy, int g_entropy_is_scalar, const float* rgb,
                           const float* opacity, const float* gt_rgb, float lambda_opacity, int n_rays, float* g_rgb, float* g_opacity,
                           ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(g_sq_err); NGP_CHECK_PTR(g_entropy); NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(gt_rgb);
    NGP_CHECK_PTR(g_rgb); NGP_CHECK_PTR(g_opacity);
    hipLaunchKernelGGL(nerf_loss_terms_bw_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream), g_sq_err,
                       g_sq_err_is_scalar, g_entropy, g_entropy_is_scalar, rgb, opacity, gt_rgb, lambda_opacity, n_rays, g_rgb, g_opacity);
    return NGP_LAUNCH_RESULT();
}

int ngp_bg_blend(const float* rgb, const float* opacity, const float* bg, int n_rays, float* rgb_out, ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(bg); NGP_CHECK_PTR(rgb_out);
    hipLaunchKernelGGL(bg_blend_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream), rgb, opacity, bg, n_rays, rgb_out);
    return NGP_LAUNCH_RESULT();
}

int ngp_bg_blend_bw(const float* g_rgb, const float* g_opacity, const float* bg, int n_rays, float* g_opacity_out, ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(g_rgb); NGP_CHECK_PTR(bg); NGP_CHECK_PTR(g_opacity_out);
    hipLaunchKernelGGL(bg_blend_bw_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream), g_rgb, g_opacity, bg, n_rays,
                       g_opacity_out);
    return NGP_LAUNCH_RESULT();
}

int ngp_sample_rays(const float* poses, const float* directions, const float* images, int n_images, int n_pixels,
                    int n, uint64_t seed, float* rays_o, float* rays_d, float* rgb, float* noise,
                    int32_t* img_idx, int32_t* pix_idx, ngp_stream_t stream) {
    if (n < 0 || n_images < 1 || n_pixels < 1) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(poses); NGP_CHECK_PTR(directions); NGP_CHECK_PTR(images); NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(rgb);
    if ((img_idx == nullptr) != (pix_idx == nullptr)) return NGP_EINVAL;
    hipLaunchKernelGGL(sample_rays_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, ngp_stream(stream), poses, directions, images,
                       n_images, n_pixels, n, (uint32_t)seed, (uint32_t)(seed >> 32), rays_o, rays_d, rgb, noise, img_idx, pix_idx);
    return NGP_LAUNCH_RESULT();
}

int ngp_sum_slices_f16(const ngp_half* own, const ngp_half* stage, int world, int rank, int64_t count, ngp_half* out, ngp_stream_t stream) {
    if (world < 1 || rank < 0 || rank >= world || count < 0 || (count & 7)) return NGP_EINVAL;
    if (count == 0) return 0;
    NGP_CHECK_PTR(own); NGP_CHECK_PTR(out);
    if (world > 1) NGP_CHECK_PTR(stage);
    if ((reinterpret_cast<uintptr_t>(own) | reinterpret_cast<uintptr_t>(stage) | reinterpret_cast<uintptr_t>(out)) & 15) return NGP_EINVAL;
    const long long n8 = count / 8;
    hipLaunchKernelGGL(sum_slices_kernel, dim3(ngp_div_up(n8, 256)), dim3(256), 0, ngp_stream(stream),
                       reinterpret_cast<const uint4*>(own), reinterpret_cast<const uint4*>(stage), world, rank, n8, reinterpret_cast<uint4*>(out));
    return NGP_LAUNCH_RESULT();
}

}  // extern "C"
