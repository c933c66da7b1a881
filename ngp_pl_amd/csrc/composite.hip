// Volumetric compositing (train fwd/bwd, test fwd) and distortion loss for gfx950.
//
// Semantics: /root/reference/models/csrc/volumerendering.cu and losses.cu (cited per kernel).
// The training kernels run ONE WAVE PER RAY with the samples on the lanes: transmittance and the
// running sums are wave scans, so products/sums associate differently from the reference's
// serial loop (differences of a few 1e-7 relative; tests bound composited outputs at 1e-5 abs)
// and the fast exponential is __expf as in the reference.  Outputs are fully written by the
// kernels (no host-side zero fill).  Test-time compositing and the distortion loss keep the
// thread-per-ray form.
#pragma clang fp contract(off)

#include "ngp_common.h"
#include "loss_common.h"

namespace {

// ---- wave64 scan helpers -----------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(v, o, 64); if (lane >= o) v *= u; }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
    return v;
}

// Per-sample transmittance state of one 64-sample chunk of a ray (volumerendering.cu:28-43):
// T_before/T_after of every sample by an inclusive product scan of (1 - alpha) across the wave,
// carried from chunk to chunk.  A sample is composited iff the transmittance BEFORE it is still
// above the threshold (the sample that crosses it is composited, then the ray stops).
//
// `live` is decided by POSITION relative to the chunk's first stop lane, not by each lane's own T_before: with the two-round
// forward (csrc/stepper.hip) the samples behind a ray's stop were never evaluated, so their sigma slots hold whatever an
// earlier step or the allocator left there.  For sane values the two rules agree (T is non-increasing: every lane up to the
// first one with T_after <= thr has T_before > thr, every later one does not), but a stale NEGATIVE sigma has 1 - a > 1 and
// could lift a later lane's T_before back above the threshold -- garbage weights, and gradients, for a ray that had stopped.
// Lanes up to the stop only look back (prefix scans), so nothing behind the stop can reach them.
struct ChunkT { float a, T_before, T_after; bool live; uint64_t stop; };
__device__ __forceinline__ ChunkT chunk_transmittance(float sigma, float delta, bool valid, float T_carry, float thr, int lane) {
    ChunkT c;
    c.a = valid ? 1.0f - __expf(-sigma * delta) : 0.0f;
    const float incl = wave_incl_prod(1.0f - c.a, lane);
    const float excl = __shfl_up(incl, 1, 64);
    c.T_before = T_carry * (lane == 0 ? 1.0f : excl);
    c.T_after = T_carry * incl;
    c.stop = __ballot(valid && c.T_after <= thr);
    const int first_stop = c.stop ? __ffsll((long long)c.stop) - 1 : 64;
    c.live = valid && lane <= first_stop && (c.T_before > thr);
    return c;
}

// volumerendering.cu:20-44.  One wave per ray: samples on lanes (coalesced streams), front-to-back
// order kept by the scans.  (The reference runs one THREAD per ray: 8192 serial chains of
// dependent strided loads, 70-90 us on MI355X for a 300 k-sample batch.)
//
// LOSS (ngp_composite_train_fw_loss): the wave also evaluates its ray's loss terms and backward seeds; one small
// kernel behind it (composite_fw_tail_kernel) turns the per-row live-sample counts into exclusive offsets and adds
// the per-row loss terms in row order -- ngp_active_scan and ngp_nerf_loss in one launch instead of two.  (Doing
// that in the last workgroup of THIS kernel behind a ticket was measured: the device-scope fences of 2048 workgroups
// cost 266 us and slowed the concurrent march by 70 %.)
struct FwTail {
    const float* gt; const float* bg; float lambda_o, grad_scale;
    float* loss; float* sq_err; float* dL_drgb; float* dL_dopacity;
    int32_t* n_active; float* row_loss; float* row_sq;
    int32_t* n_active_host;          // optional: the same count into pinned host memory (the stepper's live-fraction estimate)
    float* rgb_out;                  // optional (render()'s training branch): rgb + bg (1 - opacity), rendering.py:153-161 (bg NULL: black)
};

__device__ __forceinline__ void composite_fw_ray(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                 const float* __restrict__ deltas, const float* __restrict__ ts,
                                                 const int64_t* __restrict__ rays_a, float T_threshold, int n, int lane,
                                                 int64_t* __restrict__ total_samples, float* __restrict__ opacity,
                                                 float* __restrict__ depth, float* __restrict__ rgb, float* __restrict__ ws,
                                                 int32_t* __restrict__ n_active_per_ray, const FwTail* tail, int n_rays) {
    const int64_t ray_idx = rays_a[3 * (size_t)n];
    const int64_t start = rays_a[3 * (size_t)n + 1];
    const int N = (int)rays_a[3 * (size_t)n + 2];
    float R = 0.f, G = 0.f, B = 0.f, D = 0.f, O = 0.f, T_carry = 1.0f;
    int samples = 0;
    bool stopped = false;
    int base = 0;
    for (; base < N && !stopped; base += 64) {
        const int k = base + lane;
        const bool valid = k < N;
        const size_t s = (size_t)start + (valid ? k : 0);
        const float sigma = valid ? sigmas[s] : 0.f, delta = valid ? deltas[s] : 0.f;
        const ChunkT c = chunk_transmittance(sigma, delta, valid, T_carry, T_threshold, lane);
        const float w = c.live ? c.a * c.T_before : 0.0f;
        if (valid) ws[s] = w;
        if (c.live) {
            R += w * rgbs[3 * s]; G += w * rgbs[3 * s + 1]; B += w * rgbs[3 * s + 2];
            D += w * ts[s]; O += w;
        }
        samples += __popcll(__ballot(c.live && c.T_after > T_threshold));
        stopped = c.stop != 0ull;
        T_carry = __shfl(c.T_after, 63, 64);
    }
    for (int k = base + lane; k < N; k += 64) ws[(size_t)start + k] = 0.0f;   // chunks past the stop
    R = ngp_wave_sum(R); G = ngp_wave_sum(G); B = ngp_wave_sum(B); D = ngp_wave_sum(D); O = ngp_wave_sum(O);
    if (lane == 0) {
        rgb[3 * ray_idx] = R; rgb[3 * ray_idx + 1] = G; rgb[3 * ray_idx + 2] = B;
        depth[ray_idx] = D; opacity[ray_idx] = O;
        total_samples[ray_idx] = samples;
        if (n_active_per_ray) n_active_per_ray[n] = min(N, samples + 1);   // samples that can carry gradient (row order)
        if (tail && tail->gt) {
            const float c[3] = {R, G, B};
            const float g[3] = {tail->gt[3 * ray_idx], tail->gt[3 * ray_idx + 1], tail->gt[3 * ray_idx + 2]};
            float d_rgb[3], d_o, l = 0.f, se = 0.f;
            nerf_loss_ray(O, c, g, tail->bg, tail->lambda_o, tail->grad_scale, 1.0f / (float)n_rays, 1.0f / (3.0f * (float)n_rays),
                          d_rgb, d_o, l, se);
            tail->dL_drgb[3 * ray_idx] = d_rgb[0]; tail->dL_drgb[3 * ray_idx + 1] = d_rgb[1]; tail->dL_drgb[3 * ray_idx + 2] = d_rgb[2];
            tail->dL_dopacity[ray_idx] = d_o;
            tail->row_loss[n] = l; tail->row_sq[n] = se;
        }
        if (tail && tail->rgb_out) {                     // bg_blend_kernel's arithmetic (optim.hip), in the composite's own epilogue
            const float tr = 1.0f - O;
            const float c[3] = {R, G, B};
#pragma unroll
            for (int k = 0; k < 3; ++k) tail->rgb_out[3 * ray_idx + k] = tail->bg ? fmaf(tail->bg[k], tr, c[k]) : c[k];
        }
    }
}

// One workgroup of 1024 threads, tiles of 8192 rows, 8 consecutive rows per thread (vector loads, all in flight
// together): counts -> exclusive offsets in place, total -> *n_active; loss terms added in row order.
constexpr int TAIL_ITEMS = 8;
__global__ void __launch_bounds__(1024)
composite_fw_tail_kernel(FwTail t, int32_t* __restrict__ n_act, int n_rays) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    __shared__ float s_l[16], s_e[16];
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    float l = 0.f, se = 0.f;
    for (int base = 0; base < n_rays; base += 1024 * TAIL_ITEMS) {
        const int i0 = base + TAIL_ITEMS * tid;
        int v[TAIL_ITEMS]; float rl[TAIL_ITEMS], rs[TAIL_ITEMS];
        if (i0 + TAIL_ITEMS <= n_rays) {
#pragma unroll
            for (int q = 0; q < TAIL_ITEMS / 4; ++q) {
                const int4 a = *reinterpret_cast<const int4*>(n_act + i0 + 4 * q);
                const float4 b = *reinterpret_cast<const float4*>(t.row_loss + i0 + 4 * q);
                const float4 c = *reinterpret_cast<const float4*>(t.row_sq + i0 + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
                rl[4 * q] = b.x; rl[4 * q + 1] = b.y; rl[4 * q + 2] = b.z; rl[4 * q + 3] = b.w;
                rs[4 * q] = c.x; rs[4 * q + 1] = c.y; rs[4 * q + 2] = c.z; rs[4 * q + 3] = c.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < TAIL_ITEMS; ++k) {
                const bool in = i0 + k < n_rays;
                v[k] = in ? n_act[i0 + k] : 0; rl[k] = in ? t.row_loss[i0 + k] : 0.f; rs[k] = in ? t.row_sq[i0 + k] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < TAIL_ITEMS; ++k) { l += rl[k]; se += rs[k]; }
        int run = ngp_block_scan_tile<TAIL_ITEMS>(v, s_wave, &s_carry);
#pragma unroll
        for (int k = 0; k < TAIL_ITEMS; ++k) {
            if (i0 + k < n_rays) n_act[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();                               // the carry of this tile is visible to the next
    }
    l = ngp_wave_sum(l); se = ngp_wave_sum(se);
    if ((tid & 63) == 0) { s_l[tid >> 6] = l; s_e[tid >> 6] = se; }
    __syncthreads();
    if (tid == 0) {
        float tl = 0.f, te = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { tl += s_l[w]; te += s_e[w]; }
        *t.n_active = s_carry;
        if (t.n_active_host) *t.n_active_host = s_carry;
        *t.loss = tl;
        if (t.sq_err) *t.sq_err = te;
    }
}

// Two-round forward (csrc/stepper.hip): after the field has been evaluated on every ray's first first_k samples, which rays are
// still transparent there?  One wave per ray: the transmittance behind the first min(N, first_k) samples by the composite's own
// arithmetic (chunk_transmittance; first_k <= 64: one chunk); a ray that has not stopped and has samples left appends the ids
// of the rest to list_rest (one atomic per ray; the order of the list is irrelevant to the per-sample kernels it feeds).
// Exactly the samples volumerendering.cu:20-44 would go on to read are evaluated in the second round: the composite that
// follows never looks behind a ray's stop.
__global__ void __launch_bounds__(256)
composite_probe_kernel(const float* __restrict__ sigmas, const float* __restrict__ deltas, const int64_t* __restrict__ rays_a,
                       int first_k, float T_threshold, int n_rays, int32_t* __restrict__ list_rest, int32_t* __restrict__ n_rest) {
    __shared__ int s_rest[4], s_base;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    int rest = 0;
    int64_t start = 0;
    if (n < n_rays) {
        start = rays_a[3 * (size_t)n + 1];
        const int N = (int)rays_a[3 * (size_t)n + 2];
        if (N > first_k) {
            const bool valid = lane < first_k;
            const size_t s = (size_t)start + (valid ? lane : 0);
            const float sigma = valid ? sigmas[s] : 0.f, delta = valid ? deltas[s] : 0.f;
            const ChunkT c = chunk_transmittance(sigma, delta, valid, 1.0f, T_threshold, lane);
            if (c.stop == 0ull) rest = N - first_k;      // still transparent behind its first samples
        }
    }
    if (lane == 0) s_rest[wave] = rest;
    __syncthreads();
    if (threadIdx.x == 0) {                                    // one atomic per workgroup (4 rays), none if nothing continues
        const int total = (s_rest[0] + s_rest[1]) + (s_rest[2] + s_rest[3]);
        s_base = total ? atomicAdd(n_rest, total) : 0;
    }
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wave; ++w) base += s_rest[w];
    for (int k = lane; k < rest; k += 64) list_rest[base + k] = (int32_t)(start + first_k + k);
}

template <bool LOSS>
__global__ void __launch_bounds__(256)
composite_train_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                          const float* __restrict__ deltas, const float* __restrict__ ts,
                          const int64_t* __restrict__ rays_a, float T_threshold, int n_rays,
                          int64_t* __restrict__ total_samples, float* __restrict__ opacity,
                          float* __restrict__ depth, float* __restrict__ rgb, float* __restrict__ ws,
                          int32_t* __restrict__ n_active_per_ray, FwTail tail) {
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (n >= n_rays) return;
    composite_fw_ray(sigmas, rgbs, deltas, ts, rays_a, T_threshold, n, lane, total_samples, opacity, depth, rgb, ws,
                     n_active_per_ray, LOSS ? &tail : nullptr, n_rays);
}

// volumerendering.cu:106-150 (+ host pre-multiply :175), one wave per ray.  The running prefixes
// r,g,b,d and the prefix of dL/dw*w are wave scans; T is updated before use as in the reference.
// FUSED_TAIL (ngp_composite_train_bw_tail): `ray_offsets` holds the per-row COUNTS of live samples as the forward left them, not
// their exclusive prefix -- the one-workgroup scan kernel that used to sit between the composite forward and backward (13 us on the
// critical path for 32 KB of counts) is gone.  Every workgroup sums the counts of the rows in front of its own four (at most 32 KB
// from L2, 16-byte loads, all in flight: the sums are integers, so the offsets are exactly the scan's); the LAST workgroup, which
// holds the total anyway, writes n_active (+ the pinned host copy); workgroup 0 -- dispatched first, never the kernel's tail --
// adds the per-row loss terms in a fixed order.
struct BwTail { const float* row_loss; const float* row_sq; float* loss; float* sq_err; int32_t* n_active; int32_t* n_active_host;
                const float* bg; int blend; };     // blend: the seeds are w.r.t. render()'s BLENDED colour (row_loss may be NULL: no loss terms)
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool FUSED_TAIL>
__global__ void __launch_bounds__(256)
composite_train_bw_kernel(const float* __restrict__ dL_dopacity, const float* __restrict__ dL_ddepth,
                          const float* __restrict__ dL_drgb, const float* __restrict__ dL_dws,
                          const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                          const float* __restrict__ ws, const float* __restrict__ deltas,
                          const float* __restrict__ ts, const int64_t* __restrict__ rays_a,
                          const float* __restrict__ opacity, const float* __restrict__ depth,
                          const float* __restrict__ rgb, float T_threshold, int n_rays,
                          float* __restrict__ dL_dsigmas, float* __restrict__ dL_drgbs,
                          const int32_t* __restrict__ ray_offsets, int32_t* __restrict__ active_idx,
                          const float* __restrict__ xyzs, float* __restrict__ x_active, BwTail tail) {
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    int a_off = 0;
    if (FUSED_TAIL) {
        __shared__ int s_part[4];
        __shared__ float s_l[4], s_e[4];
        const int tid = threadIdx.x, wave = tid >> 6;
        const int row0 = 4 * (int)blockIdx.x;                              // rows in front of this workgroup: a multiple of 4
        // all of a thread's loads first (8 x 16 bytes cover 8192 rows: one L2 round trip, not one per trip of a loop -- the first
        // version's dependent trips added 8 us to a 12 us kernel, profiles/archive_r01_r04/r04_kernel_trace_summary.txt), a loop only beyond that
        int4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = 4 * tid + 1024 * q;
            v[q] = i < row0 ? *reinterpret_cast<const int4*>(ray_offsets + i) : make_int4(0, 0, 0, 0);
        }
        int acc = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        for (int i = 4 * tid + 8192; i < row0; i += 1024) {
            const int4 a = *reinterpret_cast<const int4*>(ray_offsets + i);
            acc += (a.x + a.y) + (a.z + a.w);
        }
        acc = wave_sum_int(acc);
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        const int wg_prefix = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        int own[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) own[w] = (row0 + w < n_rays) ? ray_offsets[row0 + w] : 0;
        a_off = wg_prefix;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (w < wave) a_off += own[w];
        if (blockIdx.x == gridDim.x - 1 && tid == 0) {
            const int total = wg_prefix + ((own[0] + own[1]) + (own[2] + own[3]));
            *tail.n_active = total;
            if (tail.n_active_host) *tail.n_active_host = total;
        }
        if (blockIdx.x == 0 && tail.row_loss != nullptr) {                 // loss = sum of the per-row terms, fixed order
            const int per = (n_rays + 255) / 256;
            float l = 0.f, se = 0.f;
            for (int i = tid * per; i < min((tid + 1) * per, n_rays); ++i) { l += tail.row_loss[i]; se += tail.row_sq[i]; }
            l = ngp_wave_sum(l); se = ngp_wave_sum(se);
            __syncthreads();                                               // (s_part has been read by everyone)
            if (lane == 0) { s_l[wave] = l; s_e[wave] = se; }
            __syncthreads();
            if (tid == 0) {
                *tail.loss = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
                if (tail.sq_err) *tail.sq_err = (s_e[0] + s_e[1]) + (s_e[2] + s_e[3]);
            }
        }
    }
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * (size_t)n];
    const int64_t start = rays_a[3 * (size_t)n + 1];
    const int N = (int)rays_a[3 * (size_t)n + 2];
    if (N <= 0) return;
    if (!FUSED_TAIL) a_off = active_idx ? ray_offsets[n] : 0;
    const float R = rgb[3 * ray_idx], G = rgb[3 * ray_idx + 1], B = rgb[3 * ray_idx + 2];
    const float O = opacity[ray_idx], D = depth[ray_idx];
    const float gR = dL_drgb[3 * ray_idx], gG = dL_drgb[3 * ray_idx + 1], gB = dL_drgb[3 * ray_idx + 2];
    float gO = dL_dopacity ? dL_dopacity[ray_idx] : 0.f;
    if (FUSED_TAIL && tail.blend && tail.bg) {                             // through rgb + bg (1 - opacity): bg_blend_bw_kernel's arithmetic (optim.hip)
        gO = fmaf(-gR, tail.bg[0], gO); gO = fmaf(-gG, tail.bg[1], gO); gO = fmaf(-gB, tail.bg[2], gO);
    }
    const float gD = dL_ddepth[ray_idx];
    float P_total = 0.f;   // sum of dL/dw * w over the whole segment (w = 0 past the stop)
    if (dL_dws != nullptr) {
        for (int k = lane; k < N; k += 64) { const size_t s = (size_t)start + k; P_total += dL_dws[s] * ws[s]; }
        P_total = ngp_wave_sum(P_total);
    }
    float T_carry = 1.0f, r0 = 0.f, g0 = 0.f, b0 = 0.f, d0 = 0.f, P0 = 0.f;
    bool stopped = false;
    int base = 0;
    for (; base < N && !stopped; base += 64) {
        const int k = base + lane;
        const bool valid = k < N;
        const size_t s = (size_t)start + (valid ? k : 0);
        const float sigma = valid ? sigmas[s] : 0.f, delta = valid ? deltas[s] : 0.f;
        const ChunkT c = chunk_transmittance(sigma, delta, valid, T_carry, T_threshold, lane);
        const float w = c.live ? c.a * c.T_before : 0.0f;
        float cr = 0.f, cg = 0.f, cb = 0.f, tk = 0.f, gw = 0.f;
        if (c.live) { cr = rgbs[3 * s]; cg = rgbs[3 * s + 1]; cb = rgbs[3 * s + 2]; tk = ts[s]; if (dL_dws) gw = dL_dws[s]; }
        const float r = r0 + wave_incl_sum(w * cr, lane), g = g0 + wave_incl_sum(w * cg, lane);
        const float b = b0 + wave_incl_sum(w * cb, lane), d = d0 + wave_incl_sum(w * tk, lane);
        const float P = dL_dws ? P0 + wave_incl_sum(gw * w, lane) : 0.f;
        if (valid) {
            float ds = 0.f, dr = 0.f, dg = 0.f, db = 0.f;
            if (c.live) {
                const float T = c.T_after;
                dr = gR * w; dg = gG * w; db = gB * w;
                ds = delta * (gR * (cr * T - (R - r)) + gG * (cg * T - (G - g)) + gB * (cb * T - (B - b)) +
                              gO * (1 - O) + gD * (tk * T - (D - d)) + T * gw - (P_total - P));
            }
            dL_dsigmas[s] = ds; dL_drgbs[3 * s] = dr; dL_drgbs[3 * s + 1] = dg; dL_drgbs[3 * s + 2] = db;
            if (active_idx && c.live) {                                     // live samples are the first min(N, total+1) of the ray
                active_idx[a_off + k] = (int32_t)s;
                if (x_active) {                                             // their positions in the same compact order
                    float* __restrict__ xo = x_active + 3 * (size_t)(a_off + k);
                    xo[0] = xyzs[3 * s]; xo[1] = xyzs[3 * s + 1]; xo[2] = xyzs[3 * s + 2];
                }
            }
        }
        stopped = c.stop != 0ull;
        T_carry = __shfl(c.T_after, 63, 64);
        r0 = __shfl(r, 63, 64); g0 = __shfl(g, 63, 64); b0 = __shfl(b, 63, 64); d0 = __shfl(d, 63, 64); P0 = __shfl(P, 63, 64);
    }
    for (int k = base + lane; k < N; k += 64) {
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
                           int32_t* n_active_per_ray, ngp_stream_t stream) {
    if (n_rays < 0 || n_samples < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(total_samples); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    if (n_samples > 0) { NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(ws); }
    hipLaunchKernelGGL(composite_train_fw_kernel<false>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       sigmas, rgbs, deltas, ts, rays_a, T_threshold, n_rays, total_samples, opacity, depth, rgb, ws, n_active_per_ray,
                       FwTail{});
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_probe(const float* sigmas, const float* deltas, const int64_t* rays_a, int first_k, float T_threshold,
                        int n_rays, int32_t* list_rest, int32_t* n_rest, ngp_stream_t stream) {
    if (n_rays < 0 || first_k < 1 || first_k > 64) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(list_rest); NGP_CHECK_PTR(n_rest);
    hipLaunchKernelGGL(composite_probe_kernel, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       sigmas, deltas, rays_a, first_k, T_threshold, n_rays, list_rest, n_rest);
    return NGP_LAUNCH_RESULT();
}

size_t ngp_composite_train_fw_loss_workspace_bytes(int n_rays) { return n_rays < 0 ? 0 : 8 * (((size_t)n_rays + 3) & ~(size_t)3); }

int ngp_composite_train_fw_loss(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                                int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                                int32_t* ray_offsets, int32_t* n_active, const float* gt_rgb, const float* bg,
                                float lambda_opacity, float grad_scale, float* loss, float* sq_err, float* dL_drgb,
                                float* dL_dopacity, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    return ngp_composite_train_fw_loss_h(sigmas, rgbs, deltas, ts, rays_a, T_threshold, n_rays, n_samples, total_samples, opacity, depth, rgb,
                                         ws, ray_offsets, n_active, nullptr, gt_rgb, bg, lambda_opacity, grad_scale, loss, sq_err, dL_drgb,
                                         dL_dopacity, workspace, workspace_bytes, stream);
}

int ngp_composite_train_fw_loss_h(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                  const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                                  int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                                  int32_t* ray_offsets, int32_t* n_active, int32_t* n_active_host, const float* gt_rgb, const float* bg,
                                  float lambda_opacity, float grad_scale, float* loss, float* sq_err, float* dL_drgb,
                                  float* dL_dopacity, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    if (n_rays <= 0 || n_samples < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(total_samples); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    NGP_CHECK_PTR(ray_offsets); NGP_CHECK_PTR(n_active); NGP_CHECK_PTR(gt_rgb); NGP_CHECK_PTR(loss); NGP_CHECK_PTR(dL_drgb);
    NGP_CHECK_PTR(dL_dopacity); NGP_CHECK_PTR(workspace);
    if (n_samples > 0) { NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(ws); }
    if (workspace_bytes < ngp_composite_train_fw_loss_workspace_bytes(n_rays) || ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(ray_offsets)) & 15) != 0) return NGP_EINVAL;
    FwTail t{};
    t.gt = gt_rgb; t.bg = bg; t.lambda_o = lambda_opacity; t.grad_scale = grad_scale;
    t.loss = loss; t.sq_err = sq_err; t.dL_drgb = dL_drgb; t.dL_dopacity = dL_dopacity; t.n_active = n_active;
    t.n_active_host = n_active_host;
    t.row_loss = static_cast<float*>(workspace);
    t.row_sq = t.row_loss + ((n_rays + 3) & ~3);                          // keeps the float4 loads of the tail aligned
    hipLaunchKernelGGL(composite_train_fw_kernel<true>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       sigmas, rgbs, deltas, ts, rays_a, T_threshold, n_rays, total_samples, opacity, depth, rgb, ws, ray_offsets, t);
    hipLaunchKernelGGL(composite_fw_tail_kernel, dim3(1), dim3(1024), 0, ngp_stream(stream), t, ray_offsets, n_rays);
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                           const float* dL_dws, const float* sigmas, const float* rgbs, const float* ws,
                           const float* deltas, const float* ts, const int64_t* rays_a,
                           const float* opacity, const float* depth, const float* rgb, float T_threshold,
                           int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs,
                           const int32_t* ray_offsets, int32_t* active_idx, const float* xyzs, float* x_active,
                           ngp_stream_t stream) {
    if (n_rays < 0 || n_samples < 0) return NGP_EINVAL;
    if (n_rays == 0 || n_samples == 0) return 0;
    NGP_CHECK_PTR(dL_dopacity); NGP_CHECK_PTR(dL_ddepth); NGP_CHECK_PTR(dL_drgb); NGP_CHECK_PTR(sigmas);
    NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(ws); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(rays_a);
    NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(dL_dsigmas); NGP_CHECK_PTR(dL_drgbs);
    if ((ray_offsets == nullptr) != (active_idx == nullptr)) return NGP_EINVAL;
    if ((xyzs == nullptr) != (x_active == nullptr) || (x_active != nullptr && active_idx == nullptr)) return NGP_EINVAL;
    hipLaunchKernelGGL(composite_train_bw_kernel<false>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a,
                       opacity, depth, rgb, T_threshold, n_rays, dL_dsigmas, dL_drgbs, ray_offsets, active_idx, xyzs, x_active, BwTail{});
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_train_fw_loss_counts(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                       const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                                       int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                                       int32_t* ray_counts, const float* gt_rgb, const float* bg,
                                       float lambda_opacity, float grad_scale, float* dL_drgb,
                                       float* dL_dopacity, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    if (n_rays <= 0 || n_samples < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(total_samples); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    NGP_CHECK_PTR(ray_counts); NGP_CHECK_PTR(gt_rgb); NGP_CHECK_PTR(dL_drgb); NGP_CHECK_PTR(dL_dopacity); NGP_CHECK_PTR(workspace);
    if (n_samples > 0) { NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(ws); }
    if (workspace_bytes < ngp_composite_train_fw_loss_workspace_bytes(n_rays) || ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(ray_counts)) & 15) != 0) return NGP_EINVAL;
    FwTail t{};
    t.gt = gt_rgb; t.bg = bg; t.lambda_o = lambda_opacity; t.grad_scale = grad_scale;
    t.dL_drgb = dL_drgb; t.dL_dopacity = dL_dopacity;
    t.row_loss = static_cast<float*>(workspace);
    t.row_sq = t.row_loss + ((n_rays + 3) & ~3);
    hipLaunchKernelGGL(composite_train_fw_kernel<true>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       sigmas, rgbs, deltas, ts, rays_a, T_threshold, n_rays, total_samples, opacity, depth, rgb, ws, ray_counts, t);
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_train_fw_blend(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                 const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                                 int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                                 int32_t* ray_counts, const float* bg, float* rgb_out, ngp_stream_t stream) {
    if (n_rays <= 0 || n_samples < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(total_samples); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    NGP_CHECK_PTR(ray_counts);
    if (n_samples > 0) { NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(ws); }
    if ((reinterpret_cast<uintptr_t>(ray_counts) & 15) != 0) return NGP_EINVAL;
    FwTail t{};
    t.bg = bg; t.rgb_out = rgb_out;
    hipLaunchKernelGGL(composite_train_fw_kernel<true>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       sigmas, rgbs, deltas, ts, rays_a, T_threshold, n_rays, total_samples, opacity, depth, rgb, ws, ray_counts, t);
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_train_bw_render(const float* g_opacity, const float* g_depth, const float* g_rgb,
                                  const float* g_ws, const float* sigmas, const float* rgbs, const float* ws,
                                  const float* deltas, const float* ts, const int64_t* rays_a,
                                  const float* opacity, const float* depth, const float* rgb, float T_threshold,
                                  int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs,
                                  const int32_t* ray_counts, int32_t* active_idx, const float* xyzs, float* x_active,
                                  int32_t* n_active, const float* bg, ngp_stream_t stream) {
    if (n_rays <= 0 || n_samples <= 0) return NGP_EINVAL;
    NGP_CHECK_PTR(g_depth); NGP_CHECK_PTR(g_rgb); NGP_CHECK_PTR(sigmas);
    NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(ws); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(rays_a);
    NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(dL_dsigmas); NGP_CHECK_PTR(dL_drgbs);
    NGP_CHECK_PTR(ray_counts); NGP_CHECK_PTR(active_idx); NGP_CHECK_PTR(n_active);
    if ((xyzs == nullptr) != (x_active == nullptr) || (reinterpret_cast<uintptr_t>(ray_counts) & 15) != 0) return NGP_EINVAL;
    BwTail t{};
    t.n_active = n_active; t.bg = bg; t.blend = 1;
    hipLaunchKernelGGL(composite_train_bw_kernel<true>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       g_opacity, g_depth, g_rgb, g_ws, sigmas, rgbs, ws, deltas, ts, rays_a,
                       opacity, depth, rgb, T_threshold, n_rays, dL_dsigmas, dL_drgbs, ray_counts, active_idx, xyzs, x_active, t);
    return NGP_LAUNCH_RESULT();
}

int ngp_composite_train_bw_tail(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                                const float* dL_dws, const float* sigmas, const float* rgbs, const float* ws,
                                const float* deltas, const float* ts, const int64_t* rays_a,
                                const float* opacity, const float* depth, const float* rgb, float T_threshold,
                                int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs,
                                const int32_t* ray_counts, int32_t* active_idx, const float* xyzs, float* x_active,
                                int32_t* n_active, int32_t* n_active_host, float* loss, float* sq_err,
                                const void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    if (n_rays <= 0 || n_samples <= 0) return NGP_EINVAL;               // no samples: ngp_composite_train_fw_loss (its tail kernel writes the loss)
    NGP_CHECK_PTR(dL_dopacity); NGP_CHECK_PTR(dL_ddepth); NGP_CHECK_PTR(dL_drgb); NGP_CHECK_PTR(sigmas);
    NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(ws); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts); NGP_CHECK_PTR(rays_a);
    NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb); NGP_CHECK_PTR(dL_dsigmas); NGP_CHECK_PTR(dL_drgbs);
    NGP_CHECK_PTR(ray_counts); NGP_CHECK_PTR(active_idx); NGP_CHECK_PTR(n_active); NGP_CHECK_PTR(loss); NGP_CHECK_PTR(workspace);
    if ((xyzs == nullptr) != (x_active == nullptr)) return NGP_EINVAL;
    if (workspace_bytes < ngp_composite_train_fw_loss_workspace_bytes(n_rays) || (reinterpret_cast<uintptr_t>(ray_counts) & 15) != 0) return NGP_EINVAL;
    BwTail t{};
    t.row_loss = static_cast<const float*>(workspace);
    t.row_sq = t.row_loss + ((n_rays + 3) & ~3);
    t.loss = loss; t.sq_err = sq_err; t.n_active = n_active; t.n_active_host = n_active_host;
    hipLaunchKernelGGL(composite_train_bw_kernel<true>, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a,
                       opacity, depth, rgb, T_threshold, n_rays, dL_dsigmas, dL_drgbs, ray_counts, active_idx, xyzs, x_active, t);
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
