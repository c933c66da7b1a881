// Occupancy-grid ray marching, intersection and grid helpers for gfx950.
//
// Semantics follow /root/reference/models/csrc/{intersection,raymarching}.cu (cited per kernel);
// the decomposition does not: train marching is ONE march per ray that records the sample
// parameters t into a scratch row, a single-block scan that fixes the ray-ordered packing,
// and a sample-parallel coalesced expansion (the reference marches every ray twice and packs
// in atomicAdd order).
//
// Floating-point contract: the compiler may not contract (pragma below); the two places where
// nvcc's default -fmad contracts AND the result can differ are written as explicit fmaf:
//   x = fmaf(t, d, o)            (raymarching.cu:205,246,357)
//   t1 = fmaf(dt, noise, t1)     (raymarching.cu:198)
// All other mul+add pairs in the marching maths have an exact product (power-of-two factor),
// so contraction cannot change them.  The CPU oracle makes the same choice (oracle/ngp_oracle.c).
#pragma clang fp contract(off)

#include <cstdlib>
#include <chrono>
#include "ngp_common.h"

#define NGP_SQRT3 1.73205080757f

namespace {

// ------------------------------------------------------------------------------------------
// intersection (intersection.cu:5-22, 103-121)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 aabb_hit(float ox, float oy, float oz, float ix, float iy, float iz,
                                           float cx, float cy, float cz, float hx, float hy, float hz) {
    const float tminx = (cx - hx - ox) * ix, tmaxx = (cx + hx - ox) * ix;
    const float tminy = (cy - hy - oy) * iy, tmaxy = (cy + hy - oy) * iy;
    const float tminz = (cz - hz - oz) * iz, tmaxz = (cz + hz - oz) * iz;
    const float t1 = fmaxf(fmaxf(fminf(tminx, tmaxx), fminf(tminy, tmaxy)), fminf(tminz, tmaxz));
    const float t2 = fminf(fminf(fmaxf(tminx, tmaxx), fmaxf(tminy, tmaxy)), fmaxf(tminz, tmaxz));
    if (t1 > t2) return make_float2(-1.0f, -1.0f);
    return make_float2(t1, t2);
}

__device__ __forceinline__ float2 sphere_hit(float ox, float oy, float oz, float dx, float dy, float dz,
                                             float cx, float cy, float cz, float radius) {
    const float cox = ox - cx, coy = oy - cy, coz = oz - cz;
    // dot() of helper_math.h: a.x*b.x + a.y*b.y + a.z*b.z, left to right
    const float a = dx * dx + dy * dy + dz * dz;
    const float half_b = dx * cox + dy * coy + dz * coz;
    const float c = (cox * cox + coy * coy + coz * coz) - radius * radius;
    const float disc = half_b * half_b - a * c;
    if (disc < 0) return make_float2(-1.0f, -1.0f);
    const float s = sqrtf(disc);
    return make_float2((-half_b - s) / a, (-half_b + s) / a);
}

// One thread per ray walks all primitives in index order, keeps the first max_hits hits, then
// orders the row ascending by t1 with the unfilled (-1) slots first -- what torch::sort over
// the -1-initialised buffer produces (intersection.cu:69,95-97).
template <bool SPHERE>
__global__ void __launch_bounds__(256)
ray_prim_intersect_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                          const float* __restrict__ centers, const float* __restrict__ extents,
                          int n_rays, int n_prims, int max_hits,
                          int32_t* __restrict__ hit_cnt, float* __restrict__ hits_t,
                          int64_t* __restrict__ hits_idx) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float* row_t = hits_t + (size_t)r * max_hits * 2;
    int64_t* row_i = hits_idx + (size_t)r * max_hits;
    for (int k = 0; k < max_hits; ++k) { row_t[2 * k] = -1.0f; row_t[2 * k + 1] = -1.0f; row_i[k] = -1; }
    int cnt = 0;
    for (int v = 0; v < n_prims; ++v) {
        float2 t;
        if (SPHERE) t = sphere_hit(ox, oy, oz, dx, dy, dz, centers[3 * v], centers[3 * v + 1], centers[3 * v + 2], extents[v]);
        else t = aabb_hit(ox, oy, oz, ix, iy, iz, centers[3 * v], centers[3 * v + 1], centers[3 * v + 2],
                          extents[3 * v], extents[3 * v + 1], extents[3 * v + 2]);
        if (t.y > 0) {
            if (cnt < max_hits) {
                row_t[2 * cnt] = fmaxf(t.x, 0.0f);
                row_t[2 * cnt + 1] = t.y;
                row_i[cnt] = v;
            }
            ++cnt;
        }
    }
    hit_cnt[r] = cnt;
    if (max_hits > 1) {  // stable insertion sort by t1 (the -1 rows float to the front)
        for (int i = 1; i < max_hits; ++i) {
            const float k0 = row_t[2 * i], k1 = row_t[2 * i + 1];
            const int64_t ki = row_i[i];
            int j = i - 1;
            while (j >= 0 && row_t[2 * j] > k0) {
                row_t[2 * j + 2] = row_t[2 * j]; row_t[2 * j + 3] = row_t[2 * j + 1]; row_i[j + 1] = row_i[j];
                --j;
            }
            row_t[2 * j + 2] = k0; row_t[2 * j + 3] = k1; row_i[j + 1] = ki;
        }
    }
}

// Hot path: one box, one hit, near clamp fused (rendering.py:27-29).
__global__ void __launch_bounds__(256)
ray_aabb_near_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                     const float* __restrict__ center, const float* __restrict__ half_size,
                     float near_distance, int n_rays, float* __restrict__ hits_t,
                     float* __restrict__ noise, uint32_t seed_lo, uint32_t seed_hi) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    if (noise) {          // the marcher's jitter (custom_functions.py:83 torch.rand_like): counter-based, keyed by (seed, ray)
        noise[r] = (float)(ngp_pcg_hash(ngp_rng_key(seed_lo, seed_hi, (uint32_t)r)) >> 8) * (1.0f / 16777216.0f);      // [0,1), 24 bits like torch.rand
    }
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float ix = 1.0f / rays_d[3 * r], iy = 1.0f / rays_d[3 * r + 1], iz = 1.0f / rays_d[3 * r + 2];
    const float2 t = aabb_hit(ox, oy, oz, ix, iy, iz, center[0], center[1], center[2],
                              half_size[0], half_size[1], half_size[2]);
    float t1 = -1.0f, t2 = -1.0f;
    if (t.y > 0) { t1 = fmaxf(t.x, 0.0f); t2 = t.y; }
    if (t1 >= 0 && t1 < near_distance) t1 = near_distance;
    reinterpret_cast<float2*>(hits_t)[r] = make_float2(t1, t2);
}

// ------------------------------------------------------------------------------------------
// Morton / packbits (raymarching.cu:62-161)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
morton3D_kernel(const int32_t* __restrict__ coords, int n, int32_t* __restrict__ indices) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    indices[i] = (int32_t)ngp_morton3D((uint32_t)coords[3 * i], (uint32_t)coords[3 * i + 1], (uint32_t)coords[3 * i + 2]);
}

__global__ void __launch_bounds__(256)
morton3D_invert_kernel(const int32_t* __restrict__ indices, int n, int32_t* __restrict__ coords) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t ind = indices[i];   // arithmetic shifts on int, as the reference (raymarching.cu:97-100)
    coords[3 * i] = (int32_t)ngp_compact_bits((uint32_t)(ind >> 0));
    coords[3 * i + 1] = (int32_t)ngp_compact_bits((uint32_t)(ind >> 1));
    coords[3 * i + 2] = (int32_t)ngp_compact_bits((uint32_t)(ind >> 2));
}

// 8 floats -> 1 byte per thread; the 32-byte read per lane is two dwordx4 loads, coalesced
// across the wave (2 KiB per wave-instruction pair).
template <typename T>
__global__ void __launch_bounds__(256)
packbits_kernel(const T* __restrict__ grid, int n_bytes, float thr, const float* __restrict__ stats,
                uint8_t* __restrict__ bitfield) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_bytes) return;
    if (stats != nullptr) {  // threshold = min(mean of grid>0, thr)  (networks.py:266-268)
        const float mean = stats[0] / stats[1];
        thr = (thr < mean) ? thr : mean;   // python min(mean, thr): NaN mean (no cell > 0) stays NaN -> all bits 0
    }
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) bits |= ((float)grid[(size_t)8 * n + i] > thr) ? (1u << i) : 0u;
    bitfield[n] = (uint8_t)bits;
}

// The masked mean decides the occupancy threshold (min(mean, thr), networks.py:266-268); at the
// start of training every cell's density is within a few ulp of the mean, so the SUM must not
// depend on the arrival order of atomics: workgroups park their partial sums in a library-owned
// scratch and the last one to finish (self-resetting ticket) adds them in workgroup order.
// Like ngp_nerf_loss: do not run two of these launches concurrently on different streams.
constexpr int GRID_UPDATE_BLOCKS = 256;      // every workgroup pays a fence + ticket at the end: 2048 of them took 73 us, 512 20 us
__device__ unsigned int g_grid_ticket = 0;
__device__ float g_grid_partial[2 * GRID_UPDATE_BLOCKS];

__global__ void __launch_bounds__(256)
density_grid_update_kernel(float* __restrict__ grid, const float* __restrict__ tmp,
                           const float* __restrict__ decay_grid, float decay, int n,
                           float* __restrict__ stats) {
    float sum = 0.f, cnt = 0.f;
    // 16 bytes per lane and stream (three streams, 24 MB for a 128^3 grid: 25 us with 4-byte accesses); the scalar loop takes the
    // tail of a cell count that is not a multiple of 4
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int n4 = n >> 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const f32x4 g = reinterpret_cast<const f32x4*>(grid)[i];
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tmp) + i);
        f32x4 dk = {decay, decay, decay, decay};
        if (decay_grid) dk = reinterpret_cast<const f32x4*>(decay_grid)[i];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (g[e] < 0) ? g[e] : fmaxf(g[e] * dk[e], t[e]);
            if (v[e] > 0) { sum += v[e]; cnt += 1.f; }
        }
        reinterpret_cast<f32x4*>(grid)[i] = v;
    }
    for (int i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float g = grid[i];
        const float dk = decay_grid ? decay_grid[i] : decay;
        const float v = (g < 0) ? g : fmaxf(g * dk, tmp[i]);
        grid[i] = v;
        if (v > 0) { sum += v; cnt += 1.f; }
    }
    sum = ngp_wave_sum(sum); cnt = ngp_wave_sum(cnt);
    __shared__ float s_sum[4], s_cnt[4];
    __shared__ bool s_last;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = sum; s_cnt[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        g_grid_partial[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        g_grid_partial[2 * blockIdx.x + 1] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        __threadfence();
        s_last = atomicInc(&g_grid_ticket, gridDim.x - 1) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        __threadfence();
        float a = 0.f, b = 0.f;
        for (int k = threadIdx.x; k < (int)gridDim.x; k += 64) {      // fixed order: lane k sums workgroups k, k+64, ...
            a += __builtin_nontemporal_load(&g_grid_partial[2 * k]);
            b += __builtin_nontemporal_load(&g_grid_partial[2 * k + 1]);
        }
        a = ngp_wave_sum(a); b = ngp_wave_sum(b);
        if (threadIdx.x == 0) { stats[0] += a; stats[1] += b; }        // stats keeps its "caller zeroes, call adds" contract
    }
}

// ------------------------------------------------------------------------------------------
// marching core (raymarching.cu:7-32, 204-234)
// ------------------------------------------------------------------------------------------
struct MarchParams {
    const uint8_t* __restrict__ bitfield;
    int cascades;
    int grid_size;
    float scale;        // scene scale: bound of the finest cascade is min(2^(mip-1), scale)
    float esf;          // exp_step_factor
    float dt_lo;        // SQRT3 / max_samples
    float dt_hi;        // SQRT3 * 2 * scale_for_dt / grid_size
    float bound0, bound0_inv;   // mip 0: fminf(scalbnf(1, -1), scale) and its reciprocal (SIMPLE path)
    int simple;         // cascades == 1 && esf == 0
};

__device__ __forceinline__ float calc_dt(float t, const MarchParams& p) {
    return fmaxf(p.dt_lo, fminf(t * p.esf, p.dt_hi));
}

// Termination guards.  Every marching loop below ends because t strictly increases; that only holds while a step is not
// absorbed by rounding (t + dt == t once ulp(t) > 2 dt, i.e. t >~ 3e4 for the smallest step) and the far hit is finite --
// true for every ray an AABB / sphere intersection produces, not for arbitrary caller input (the reference's loops,
// raymarching.cu:225-232, spin for ever there).  A tripped guard ends the RAY (never the kernel's other rays), and counts
// itself in g_march_guard, which ngp_march_guard_read() hands to the host: [0] a skip whose step would be absorbed,
// [1] the wave-per-ray tile cap, [2] a serial loop's iteration cap (train / test / frame loop), [3] 1 + index of the last ray
// that ran into [1] (diagnostics).
__device__ unsigned int g_march_guard[4];
__device__ float g_march_guard_first[12];       // the first probe that ran into [0]: t, t_target, tx, ty, tz, o (3), d (3), dt_lo
#ifdef NGP_RENDER_TIMING
// diagnostics build (tools/build_variant.sh ... -DNGP_RENDER_TIMING; tools/render_wave_times.py): one row per wave of the frame loop's
// thread-per-ray marcher -- wall clock at entry and exit (100 MHz), the wave's longest and summed probe counts, N and the alive count,
// and the longest lane's own mix: hops of at most 8 lattice steps (a cell's diagonal), longer hops (blocks), samples
constexpr unsigned int RENDER_TIMING_ROWS = 1u << 18;
__device__ unsigned long long g_render_timing[6 * RENDER_TIMING_ROWS];
__device__ unsigned int g_render_timing_n;
#endif
constexpr int MARCH_TILE_CAP = 1 << 14;          // 2^20 lattice points per ray; the longest legitimate ray (scale 64) has 2^17
constexpr int MARCH_ITER_CAP = 1 << 20;

struct Ray {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
};

// Evaluates the cell containing the sample at parameter t.  Returns whether it is occupied,
// the sample position, dt, and (when empty) the next t on the step lattice past the cell.
// SIMPLE = one cascade and exp_step_factor 0 (the Synthetic-NeRF setting): then mip is 0 for every
// sample (min(cascades-1, .) = 0), mip_bound is the constant min(0.5, scale) and
// dt = max(dt_lo, min(t*0, dt_hi)) = dt_lo for every t >= 0, so the two frexp, the scalbn, the
// division and the clamp leave the dependent chain; every value that remains is computed by
// the same operations as in the general case.
// block_any (optional, LDS; SIMPLE only): bit b = "some cell of the 8^3 block b = idx >> 9 (512 Morton-consecutive cells) is
// occupied".  In a block without one the serial walk hops cell by cell (up to 22 hops of ~110 instructions) and emits nothing; where it
// comes out does not depend on those hops: every hop lands on L(tau) = the first lattice point at or behind tau, tau = the float
// evaluation of the time the ray leaves the current cell, and the LAST hop inside the block evaluates the time the ray leaves the
// block -- the same plane whichever cell it is taken from.  Two evaluations of that time, from different points of the ray, differ
// by rounding only: each is within d = |t| 2^-23 + (tau - t) 2^-22 + 2^-23 max|1/d_axis| of the real crossing (x = fma(t, d, o)
// is within half an ulp of a value below 1, the plane is exact, one subtraction, one product with the rounded reciprocal, one sum).
// So when no lattice point lies within `hop_slack` >= 2 d (x4 for safety) of the block's tau evaluated HERE, L(tau) is the point the
// cell-by-cell walk reaches, and the hop goes there at once; lattice points that close to the boundary (about 1% of the hops) take
// the cells one by one as before.  Points between here and there are inside the block by more than the rounding of their own
// position, so none of them can have been attributed to a neighbouring (occupied) block.  A set bit reads the cell as before.
template <bool SIMPLE>
__device__ __forceinline__ bool march_probe(const Ray& ray, const MarchParams& p, float t,
                                            float& x, float& y, float& z, float& dt, float& t_next, int* steps = nullptr,
                                            const uint32_t* block_any = nullptr, float hop_slack = 0.0f) {
    x = fmaf(t, ray.dx, ray.ox); y = fmaf(t, ray.dy, ray.oy); z = fmaf(t, ray.dz, ray.oz);
    const float G = (float)p.grid_size;
    int mip = 0;
    if (SIMPLE) {
        dt = p.dt_lo;
    } else {
        dt = calc_dt(t, p);
        int e_pos; frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e_pos);
        int e_dt; frexpf(dt * G, &e_dt);
        const int mip_pos = min(p.cascades - 1, max(0, e_pos + 1));
        const int mip_dt = min(p.cascades - 1, max(0, e_dt));
        mip = max(mip_pos, mip_dt);
    }
    const float mip_bound = SIMPLE ? p.bound0 : fminf(scalbnf(1.0f, mip - 1), p.scale);
    const float mip_bound_inv = SIMPLE ? p.bound0_inv : 1 / mip_bound;
    const float gm1 = G - 1.0f;
    const int nx = (int)fmaxf(0.0f, fminf(0.5f * (x * mip_bound_inv + 1) * G, gm1));
    const int ny = (int)fmaxf(0.0f, fminf(0.5f * (y * mip_bound_inv + 1) * G, gm1));
    const int nz = (int)fmaxf(0.0f, fminf(0.5f * (z * mip_bound_inv + 1) * G, gm1));
    const uint32_t g3 = (uint32_t)(p.grid_size * p.grid_size * p.grid_size);
    const uint32_t idx = (uint32_t)mip * g3 + ngp_morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    bool occ = true;
    if (SIMPLE && block_any != nullptr) {
        occ = (block_any[idx >> 14] >> ((idx >> 9) & 31)) & 1;
        if (!occ) {
            const float g8 = 8.0f / G;
            const float ux = ((((nx >> 3) + 0.5f + 0.5f * copysignf(1.0f, ray.dx)) * g8 * 2 - 1) * mip_bound - x) * ray.ix;
            const float uy = ((((ny >> 3) + 0.5f + 0.5f * copysignf(1.0f, ray.dy)) * g8 * 2 - 1) * mip_bound - y) * ray.iy;
            const float uz = ((((nz >> 3) + 0.5f + 0.5f * copysignf(1.0f, ray.dz)) * g8 * 2 - 1) * mip_bound - z) * ray.iz;
            const float tau = t + fmaxf(0.0f, fminf(ux, fminf(uy, uz)));
            // L(tau) and its predecessor without the chain of adds (a wave pays every lane's longest loop): t and tau in one binade,
            // where the constant step adds the same number of ulps every time (lattice_tile_const_dt has the argument); a hop that
            // would leave the binade, a step that ties, a t below 2 dt take the cells one by one
            const uint32_t tb = __float_as_uint(t), ub = __float_as_uint(tau), db = __float_as_uint(p.dt_lo);
            const int sh = (int)(tb >> 23) - (int)(db >> 23);                  // (signs are 0: t > 0, dt > 0)
            if (tau - t < 0.25f && (ub >> 23) == (tb >> 23) && sh >= 1 && sh <= 23 && (db >> 23) != 0u && (tb >> 23) != 0u) {
                const uint32_t md = (db & 0x7fffffu) | 0x800000u;
                const uint32_t rem = md & ((1u << sh) - 1u), half = 1u << (sh - 1);
                const uint32_t delta = (md >> sh) + (rem > half ? 1u : 0u);
                const uint32_t diff = (ub & 0x7fffffu) - (tb & 0x7fffffu);           // tau >= t, same binade
                uint32_t k = (uint32_t)((float)diff * __builtin_amdgcn_rcpf((float)delta));
                if (k * delta < diff) ++k;                                            // k = the smallest count with k delta >= diff, at least 1:
                if (k * delta < diff) ++k;                                            // the quotient above is within one of it
                if (k > 1u && (k - 1u) * delta >= diff) --k;
                if (k == 0u) k = 1u;
                const uint32_t mk = (tb & 0x7fffffu) + k * delta;
                if (rem != half && delta != 0u && mk < 0x800000u) {
                    const float tt = __uint_as_float((tb & 0xff800000u) | mk), prev = __uint_as_float((tb & 0xff800000u) | (mk - delta));
                    if (tt - tau > hop_slack && tau - prev > hop_slack) {
                        t_next = tt;
                        if (steps) *steps = (int)k;
                        return false;
                    }
                }
            }
        }
    }
    if (occ) occ = (p.bitfield[idx >> 3] >> (idx & 7)) & 1;
    if (!occ) {
        const float ginv = 1.0f / G;
        const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, ray.dx)) * ginv * 2 - 1) * mip_bound - x) * ray.ix;
        const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, ray.dy)) * ginv * 2 - 1) * mip_bound - y) * ray.iy;
        const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, ray.dz)) * ginv * 2 - 1) * mip_bound - z) * ray.iz;
        const float t_target = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        float tt = t;
        int k = 0;                                   // elements of the ray's t sequence the skip advances by (>= 1)
        if (t_target + p.dt_lo == t_target) {        // the smallest step no longer moves t anywhere up to the target (false for NaN): end the ray
            if (atomicAdd(&g_march_guard[0], 1u) == 0u) {
                float* q = g_march_guard_first;
                q[0] = t; q[1] = t_target; q[2] = tx; q[3] = ty; q[4] = tz; q[5] = ray.ox; q[6] = ray.oy; q[7] = ray.oz;
                q[8] = ray.dx; q[9] = ray.dy; q[10] = ray.dz; q[11] = p.dt_lo;
            }
            tt = __builtin_inff(); k = 1 << 20;
        } else if (SIMPLE) { do { tt += p.dt_lo; ++k; } while (tt < t_target); }
        else { do { tt += calc_dt(tt, p); ++k; } while (tt < t_target); }
        t_next = tt;
        if (steps) *steps = k;
    }
    return occ;
}

__device__ __forceinline__ Ray load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, size_t r) {
    Ray ray;
    ray.ox = rays_o[3 * r]; ray.oy = rays_o[3 * r + 1]; ray.oz = rays_o[3 * r + 2];
    ray.dx = rays_d[3 * r]; ray.dy = rays_d[3 * r + 1]; ray.dz = rays_d[3 * r + 2];
    ray.ix = 1.0f / ray.dx; ray.iy = 1.0f / ray.dy; ray.iz = 1.0f / ray.dz;
    return ray;
}

// Pass 1 of train marching (raymarching.cu:184-234): one march per ray, t of every emitted sample goes to the ray's scratch row.
// The reference's loop is a chain of dependent bitfield loads with per-ray trip counts from 0 to ~600 (a serial-chain kernel, 16 rays
// per wave, took 280-330 us per 8192-ray batch: round 1; removed in round 5, tests/test_march_parallel_proto_cpu.py keeps the numpy
// prototype of the equivalence).
// T[j] = fl(T[j-1] + dt), j = 1..64, for a CONSTANT step (the Synthetic-NeRF setting), in closed form instead of a chain of
// 64 dependent adds: inside one binade [2^e, 2^(e+1)) every element is a multiple of u = 2^(e-23) and dt / u = q + r with a
// fraction r that is the same for every element, so round-to-nearest adds the SAME integer number of ulps each step:
// delta = q + (r > 1/2).  Lane i of a binade segment starting at mantissa m0 therefore holds m0 + i * delta exactly as long as
// that stays below 2^24; the element that crosses into the next binade is formed by a real float add from its predecessor
// (its rounding follows the new binade's ulp) and starts the next segment.  Returns false where the closed form does not
// hold bit for bit (r == 1/2: ties round to even and the increment depends on the mantissa's parity; t below dt; zero /
// subnormal t): the caller then runs the chain.  Wave-uniform t_start, dt; lane j receives T[j], every lane T[64] in t_end.
__device__ __forceinline__ bool lattice_tile_const_dt(float t_start, float dt, int lane, float& mine, float& t_end) {
    const uint32_t db = __float_as_uint(dt);
    const int ed = (int)((db >> 23) & 0xffu) - 127;
    const uint32_t md = (db & 0x7fffffu) | 0x800000u;
    if (((db >> 23) & 0xffu) == 0u) return false;
    uint32_t seg_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(t_start));
    int seg_j = 0;
    mine = t_start;
    while (seg_j < 64) {
        const int be = (int)((seg_bits >> 23) & 0xffu);
        const int s = (be - 127) - ed;
        if (be == 0 || be == 255 || (seg_bits >> 31) != 0u || s < 1 || s > 23) return false;
        const uint32_t rem = md & ((1u << s) - 1u), half = 1u << (s - 1);
        if (rem == half) return false;
        const uint32_t delta = (md >> s) + (rem > half ? 1u : 0u);
        const uint32_t m0 = (seg_bits & 0x7fffffu) | 0x800000u;
        const uint32_t mj = m0 + (uint32_t)(lane - seg_j) * delta;           // < 2^24 + 63 * 2^23: no wrap
        const bool in = lane >= seg_j && mj < 0x1000000u;
        if (in) mine = __uint_as_float(((uint32_t)be << 23) | (mj & 0x7fffffu));
        const int n_in = (int)__popcll(__ballot(in));                           // >= 1: the segment's first element is in its own binade
        const int jc = seg_j + n_in;
        // the element behind the segment: a real add from the segment's last element (T[64] when the tile is complete)
        const float last = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mine), jc - 1));
        const float nxt = last + dt;
        if (jc >= 64) { t_end = nxt; return true; }
        seg_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(nxt));
        seg_j = jc;
    }
    return true;
}

// ONE WAVE PER RAY, bit-identical to the serial loop.  The loop visits a subsequence of one
// fixed sequence per ray, T[0] = t1, T[j+1] = T[j] + calc_dt(T[j]): an occupied cell advances by one element, an empty
// cell by k >= 1 elements (the do-while of the skip).  Per tile of 64 elements the wave
//   1. generates the 64 elements (a chain of adds, no memory) -- lane j keeps T[j];
//   2. probes all 64 candidates in parallel (march_probe, the serial kernel's own arithmetic): occupancy bit and, for
//      empty cells, the skip length k, i.e. the successor index j + k;
//   3. follows the orbit of the tile's entry index under `successor`: chains of consecutive skips are closed on the vector
//      unit by pointer doubling (lane -> landing lane of the whole chain), then wave-uniform scalar code walks the orbit --
//      runs of occupied lanes are taken at once from the ballot mask, a chain of skips costs one cross-lane read (the
//      scalar unit is shared by the CU's four SIMDs: one read per skipped cell made the kernel SALU-bound).  Candidates the
//      serial loop jumps over are never emitted, whatever their own bit says (next to a voxel face the computed skip
//      target can reach a rounding error into the neighbour cell);
//   4. writes the visited occupied candidates' t to the ray's scratch row (coalesced, ranked by popcount).
// A skip that leaves the tile carries its landing value into the following tiles.  tools/march_parallel_proto.py is
// this algorithm in numpy against the oracle.  8192 rays = 8192 waves instead of 512 serial chains of dependent loads.
// PROLOGUE (the native stepper's march): the wave also forms its ray's hit interval and jitter -- render()'s prologue
// (rendering.py:27-29: one box, one hit, near clamp; custom_functions.py:83: the jitter draw), the arithmetic of ray_aabb_near_kernel
// -- instead of reading them from a launch of their own, and leaves them in hits_out / noise_out for whoever reads them later.
struct MarchPrologue { const float* center; const float* half_size; float near_distance; uint32_t seed_lo, seed_hi; float* hits_out; float* noise_out; };
template <bool SIMPLE, bool PROLOGUE>
__global__ void __launch_bounds__(256)
march_train_count_wave_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                              const float* __restrict__ hits_t, const float* __restrict__ noise,
                              MarchParams p, int max_samples, int n_rays,
                              int64_t* __restrict__ rays_a, float* __restrict__ t_scratch, int32_t* __restrict__ counts, MarchPrologue pro) {
    // the ray index is wave-uniform; saying so keeps the ray, its hit interval and the whole tile walk on the scalar unit
    const int r = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const Ray ray = load_ray(rays_o, rays_d, r);
    float t1, t2, jitter;
    if (PROLOGUE) {
        const float2 t = aabb_hit(ray.ox, ray.oy, ray.oz, ray.ix, ray.iy, ray.iz, pro.center[0], pro.center[1], pro.center[2],
                                  pro.half_size[0], pro.half_size[1], pro.half_size[2]);
        t1 = -1.0f; t2 = -1.0f;
        if (t.y > 0) { t1 = fmaxf(t.x, 0.0f); t2 = t.y; }
        if (t1 >= 0 && t1 < pro.near_distance) t1 = pro.near_distance;
        jitter = (float)(ngp_pcg_hash(ngp_rng_key(pro.seed_lo, pro.seed_hi, (uint32_t)r)) >> 8) * (1.0f / 16777216.0f);
        if (lane == 0) { reinterpret_cast<float2*>(pro.hits_out)[r] = make_float2(t1, t2); pro.noise_out[r] = jitter; }
    } else {
        t1 = hits_t[2 * r]; t2 = hits_t[2 * r + 1]; jitter = noise[r];
    }
    if (t1 >= 0) t1 = fmaf(calc_dt(t1, p), jitter, t1);
    float* __restrict__ row = t_scratch + (size_t)r * max_samples;
    int n = 0;
    float t_start = t1;
    float pending = -1.0f;                              // landing value of a skip that left the previous tile (< 0: none)
    bool done = !(t1 >= 0);
    int tiles = 0;
    while (!done) {
        if (++tiles > MARCH_TILE_CAP) { if (lane == 0) { atomicAdd(&g_march_guard[1], 1u); g_march_guard[3] = (unsigned)r + 1u; } break; }
        // 1. the tile's elements: closed form for the constant step (lattice_tile_const_dt), else the chain of 64 adds
        float mine = t_start, t_end = t_start;
        if (!(SIMPLE && lattice_tile_const_dt(t_start, p.dt_lo, lane, mine, t_end))) {
            float tt = t_start;
            mine = t_start;
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                mine = (lane == j) ? tt : mine;
                tt += SIMPLE ? p.dt_lo : calc_dt(tt, p);
            }
            t_end = tt;                                 // T[64]: first element of the next tile
        }
        const int nvalid = __popcll(__ballot(0 <= mine && mine < t2));        // the sequence increases: valid lanes are a prefix
        const int entry = pending >= 0 ? __popcll(__ballot(mine < pending)) : 0;
        if (entry >= 64) {                              // the carried skip jumps over the whole tile
            if (nvalid < 64) break;                     // ... and over the end of the ray
            t_start = t_end;
            continue;
        }
        // 2. all candidates at once
        float x, y, z, dt, t_next = 0.f;
        int k = 1;
        const bool occ = march_probe<SIMPLE>(ray, p, mine, x, y, z, dt, t_next, &k);
        const unsigned long long occ_mask = __ballot(occ);
        const unsigned long long empty_mask = ~occ_mask;
        // 3a. empty lanes: where does the chain of skips that starts here end?  Pointer doubling on the vector unit over
        //     packed (last empty lane of the chain << 8 | landing lane): after r rounds a lane knows the end of a chain of 2^r
        //     skips, so 6 rounds close every chain of a 64-lane tile (a skip advances by >= 1); rounds stop as soon as no lane's
        //     landing lane is empty any more (typically after 3-4).  The landing lane is occupied, or >= 64 (the chain leaves the tile).
        int chain = occ ? ((lane << 8) | 64) : ((lane << 8) | (lane + k > 64 ? 64 : lane + k));
#pragma unroll 1
        for (int round = 0; round < 6; ++round) {
            const int h = chain & 0xff;
            const bool go = h < 64 && ((empty_mask >> h) & 1ull);
            const int via = __builtin_amdgcn_ds_bpermute((h & 63) << 2, chain);       // the chain that starts at the landing lane
            if (go) chain = via;
            if (!__ballot(go)) break;
        }
        // 3b. the orbit of `entry` (wave-uniform, on the scalar unit): occupied runs come from the ballot mask, a whole chain of
        //     skips costs one cross-lane read
        unsigned long long emit = 0ull;
        float next_pending = -1.0f;
        bool finished = false;
        int v = entry;
        while (v < 64) {
            if (v >= nvalid) { finished = true; break; }
            if ((empty_mask >> v) & 1ull) {
                const int pk = __builtin_amdgcn_readlane(chain, v);                 // v is wave-uniform
                const int h = pk & 0xff;
                if (h >= 64) {                                                        // the chain's last skip leaves the tile: carry its landing value
                    next_pending = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_next), pk >> 8));
                    v = 64;
                    break;
                }
                v = h;
                continue;
            }
            const unsigned long long un = empty_mask >> v;
            const int u = un ? v + (int)__builtin_ctzll(un) : 64;                   // lanes v .. u-1 are occupied
            const int hi = u < nvalid ? u : nvalid;
            if (hi > v) {
                const int len = hi - v;
                emit |= (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << v;
            }
            if (nvalid < 64 && u >= nvalid) { finished = true; break; }             // the run (or the empty lane behind it) reaches the far hit
            v = u;                                                                    // 64: the tile ends inside a run; else the empty lane behind the run
        }
        // 4. write out, capped at max_samples like the serial loop's N_samples < max_samples
        const int room = max_samples - n;
        const bool my = (emit >> lane) & 1ull;
        const int rank = __popcll(emit & ((1ull << lane) - 1ull));
        if (my && rank < room) row[n + rank] = mine;
        const int cnt = __popcll(emit);
        n += cnt < room ? cnt : room;
        if (finished || n >= max_samples) break;
        pending = next_pending;
        if (pending >= 0 && !(pending < t2)) break;     // the skip lands beyond the far hit
        t_start = t_end;
    }
    if (lane == 0) {
        rays_a[3 * (size_t)r] = r;
        rays_a[3 * (size_t)r + 2] = n;
        if (counts) counts[r] = n;                      // the same counts as a dense i32 array: what the self-prefixing expansion sums
    }
}

// Exclusive scan of rays_a[:,2] into rays_a[:,1] in ray order; counter = {S, R}.
// Single 1024-thread workgroup, tiles of 8192 rays (8 consecutive rays per thread, all loads first).
// offs_k != nullptr (two-round forward): also the exclusive scan of min(N, first_k) into offs_k -- where every ray's first
// samples go in the compact first-round list -- and its total into counter[3].
__global__ void __launch_bounds__(1024)
march_train_scan_kernel(int64_t* __restrict__ rays_a, int n_rays, int32_t* __restrict__ counter, int first_k, int32_t* __restrict__ offs_k) {
    __shared__ int s_wave[16], s_wave_k[16];
    __shared__ int s_carry, s_carry_k;
    const int tid = threadIdx.x;
    if (tid == 0) { s_carry = 0; s_carry_k = 0; }
    __syncthreads();
    for (int base = 0; base < n_rays; base += 8192) {
        const int r0 = base + 8 * tid;
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (r0 + k < n_rays) ? (int)rays_a[3 * (size_t)(r0 + k) + 2] : 0;
        int run = ngp_block_scan_tile<8>(v, s_wave, &s_carry);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (r0 + k < n_rays) rays_a[3 * (size_t)(r0 + k) + 1] = run;
            run += v[k];
        }
        if (offs_k) {                                   // (uniform)
            int vk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) vk[k] = min(v[k], first_k);
            int run_k = ngp_block_scan_tile<8>(vk, s_wave_k, &s_carry_k);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (r0 + k < n_rays) offs_k[r0 + k] = run_k;
                run_k += vk[k];
            }
        }
        __syncthreads();
    }
    if (tid == 0) { counter[0] = s_carry; counter[1] = n_rays; if (offs_k) counter[3] = s_carry_k; }
}

// Pass 2 of train marching: expand (ray, k) -> packed sample.  One wave per ray, lanes stride
// the ray's samples: scratch reads and all four output streams are coalesced.
// first_k > 0 (two-round forward, csrc/stepper.hip): the ids of every ray's first min(N, first_k) samples are also written to
// list_k[ray * first_k + k] (-1 where the ray has fewer: a padded list of n_rays * first_k entries -- no counter, no atomics; 8192
// atomics on one address cost 28 us, one per 4-ray workgroup still 19 us), or, with the offsets offs_k the march's scan kernel
// computed, to a COMPACT list in ray order (late in training 70 % of the rays have no samples: the padded list is mostly padding).
// *n_clear (the counter of the second round's list) is cleared on the side.
__global__ void __launch_bounds__(256)
march_train_write_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                         const int64_t* __restrict__ rays_a, const float* __restrict__ t_scratch,
                         MarchParams p, int max_samples, int n_rays,
                         float* __restrict__ xyzs, float* __restrict__ dirs,
                         float* __restrict__ deltas, float* __restrict__ ts,
                         int first_k, int32_t* __restrict__ list_k, int32_t* __restrict__ n_clear,
                         const int32_t* __restrict__ offs_k, const int32_t* __restrict__ counts = nullptr,
                         int64_t* __restrict__ rays_a_out = nullptr, int32_t* __restrict__ counter = nullptr) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (first_k > 0 && blockIdx.x == 0 && threadIdx.x == 0 && n_clear) *n_clear = 0;
    const bool in = wave < n_rays;
    int64_t r, start; int n;
    if (counts != nullptr) {
        // SELF-PREFIXING form (the native stepper's march: no scan kernel in between).  Rays are packed in ray order, so ray w's first
        // sample sits at the sum of the counts of the rays in front of it: the workgroup (4 rays) sums counts[0 .. 4 blockIdx) itself --
        // 16-byte loads from a dense i32 array the L2 holds (<= 32 KB, 33 MB over the whole launch) -- and publishes the total to the
        // host (pinned counter, system scope) from the LAST workgroup as soon as it has it, before any expansion work.
        __shared__ int s_part[4];
        const int first = (int)blockIdx.x * 4;              // rays [first, first + 4) are this workgroup's
        int acc = 0;
        for (int i = (int)threadIdx.x * 4; i < first; i += 1024) {
            const int4 v = *reinterpret_cast<const int4*>(counts + i);
            acc += (v.x + v.y) + (v.z + v.w);
        }
        acc = ngp_wave_sum_i32(acc);
        if (lane == 0) s_part[threadIdx.x >> 6] = acc;
        __syncthreads();
        int base = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        const int w_in = (int)(threadIdx.x >> 6);
        int mine = 0, total = base;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (first + j < n_rays) ? counts[first + j] : 0;
            if (j < w_in) base += c;
            if (j == w_in) mine = c;
            total += c;
        }
        r = wave; start = base; n = in ? mine : 0;
        if (in && lane == 0) rays_a_out[3 * (size_t)wave + 1] = start;      // (ray id and count were written by the count kernel)
        if (counter != nullptr && first + 4 >= n_rays && threadIdx.x == 0) {
            __hip_atomic_store(counter + 1, n_rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(counter, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {
        r = in ? rays_a[3 * (size_t)wave] : 0;
        start = in ? rays_a[3 * (size_t)wave + 1] : 0;
        n = in ? (int)rays_a[3 * (size_t)wave + 2] : 0;
    }
    if (first_k > 0 && in) {
        if (offs_k) { if (lane < min(n, first_k)) list_k[offs_k[wave] + lane] = (int32_t)(start + lane); }       // compact, in ray order
        else if (lane < first_k) list_k[(size_t)wave * first_k + lane] = lane < n ? (int32_t)(start + lane) : -1;
    }
    if (!in) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float* __restrict__ row = t_scratch + (size_t)r * max_samples;
    for (int k = lane; k < n; k += 64) {
        const float t = row[k];
        const size_t s = (size_t)start + k;
        xyzs[3 * s] = fmaf(t, dx, ox); xyzs[3 * s + 1] = fmaf(t, dy, oy); xyzs[3 * s + 2] = fmaf(t, dz, oz);
        dirs[3 * s] = dx; dirs[3 * s + 1] = dy; dirs[3 * s + 2] = dz;
        ts[s] = t;
        deltas[s] = calc_dt(t, p);
    }
}

// Test-time marching (raymarching.cu:353-403).  Dense (n_alive, n_samples) outputs are fully
// written here (the reference zero-fills them on the host first).
template <bool SIMPLE>
__global__ void __launch_bounds__(64)
march_test_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                  float* __restrict__ hits_t, const int64_t* __restrict__ alive,
                  MarchParams p, int n_samples, int n_alive,
                  float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                  float* __restrict__ ts, int32_t* __restrict__ n_eff) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const size_t r = (size_t)alive[n];
    const Ray ray = load_ray(rays_o, rays_d, r);
    float t = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    const size_t base = (size_t)n * n_samples;
    int s = 0, iters = 0;
    float t_resume = t;
    while (t < t2 && s < n_samples) {
        if (++iters > MARCH_ITER_CAP) { atomicAdd(&g_march_guard[2], 1u); t_resume = t2; break; }
        float x, y, z, dt, t_next;
        if (march_probe<SIMPLE>(ray, p, t, x, y, z, dt, t_next)) {
            const size_t o = base + s;
            xyzs[3 * o] = x; xyzs[3 * o + 1] = y; xyzs[3 * o + 2] = z;
            dirs[3 * o] = ray.dx; dirs[3 * o + 1] = ray.dy; dirs[3 * o + 2] = ray.dz;
            ts[o] = t; deltas[o] = dt;
            t += dt; ++s;
            t_resume = t;           // raymarching.cu:390: stored after every emitted sample only
        } else {
            t = t_next;
        }
    }
    if (s > 0) hits_t[2 * r] = t_resume;
    n_eff[n] = s;
    for (int k = s; k < n_samples; ++k) {
        const size_t o = base + k;
        xyzs[3 * o] = 0.f; xyzs[3 * o + 1] = 0.f; xyzs[3 * o + 2] = 0.f;
        dirs[3 * o] = 0.f; dirs[3 * o + 1] = 0.f; dirs[3 * o + 2] = 0.f;
        ts[o] = 0.f; deltas[o] = 0.f;
    }
}

// Alive-ray compaction of the test-time loop (`alive_indices[alive_indices>=0]`,
// rendering.py:105) without torch's nonzero(): wave ballot + one atomic per wave.  The order of
// the survivors is not preserved (rays are independent, nothing depends on it).  Also sums the
// rays' N_eff into total[0] (rendering.py:88).
__global__ void __launch_bounds__(256)
compact_alive_kernel(const int64_t* __restrict__ alive_in, const int32_t* __restrict__ n_eff, int n,
                     int64_t* __restrict__ alive_out, int32_t* __restrict__ count, int64_t* __restrict__ total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t a = (i < n) ? alive_in[i] : -1;
    const bool keep = a >= 0;
    const unsigned long long m = __ballot(keep);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0, 64);
    if (keep) alive_out[base + __popcll(m & ((1ull << lane) - 1ull))] = a;
    if (total != nullptr) {
        int e = (i < n && n_eff != nullptr) ? n_eff[i] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
        if (lane == 0 && e) atomicAdd(reinterpret_cast<unsigned long long*>(total), (unsigned long long)e);
    }
}

MarchParams make_march_params(const uint8_t* bitfield, int cascades, int grid_size, float scale,
                              float scale_for_dt, float esf, int max_samples) {
    MarchParams p;
    p.bitfield = bitfield; p.cascades = cascades; p.grid_size = grid_size; p.scale = scale; p.esf = esf;
    p.dt_lo = NGP_SQRT3 / max_samples;                 // raymarching.cu:12, float / int
    p.dt_hi = NGP_SQRT3 * 2 * scale_for_dt / grid_size;
    p.bound0 = fminf(scalbnf(1.0f, -1), scale);
    p.bound0_inv = 1 / p.bound0;
    p.simple = (cascades == 1 && esf == 0.0f) ? 1 : 0;
    return p;
}


// ------------------------------------------------------------------------------------------
// Device-driven test-time frame loop (rendering.py:46-118).  Per iteration: march -> hash grid
// -> MLPs -> composite+compaction; the alive count, N_samples and the batch size are device
// state (RenderPlan), so no kernel waits for a host read.
// ------------------------------------------------------------------------------------------
struct RenderPlan {            // one per loop iteration
    int32_t n_alive_raw;       // survivors of the previous iteration (atomically counted)
    int32_t samples_done;      // `samples` (rendering.py:71) before this iteration
    int32_t n_alive;           // rays marched this iteration (0 once samples_done >= max_samples)
    int32_t n_step;            // N_samples of this iteration (rendering.py:69)
    int32_t m;                 // n_alive * n_step sample slots
    int32_t blocks_done;       // composite workgroups that have added their survivors
    int32_t pad[2];
};
constexpr int RENDER_RETIRE = 1 << 16;      // n_eff flag: drop the ray after compositing its samples
constexpr int RENDER_MAX_ITERS = 2048;
constexpr int RENDER_RING = 4;

// block_any (RENDER_MASK_WORDS words, or NULL): the "any cell of the 8^3 block occupied" bits the marcher keeps in LDS (march_probe);
// built here from the bitfield the frame was called with, by the first 16 workgroups (one thread per block: 64 bytes of bitfield).
constexpr int RENDER_MASK_WORDS = 128;           // (128 / 8)^3 blocks
__global__ void __launch_bounds__(256)
render_begin_kernel(const float* __restrict__ hits_in, int n_rays, float* __restrict__ hits,
                    int32_t* __restrict__ alive, int32_t* __restrict__ emitted, float* __restrict__ opacity, float* __restrict__ depth,
                    float* __restrict__ rgb, RenderPlan* __restrict__ plan, unsigned long long* __restrict__ total,
                    const uint8_t* __restrict__ bitfield, uint32_t* __restrict__ block_any, int hits_pairs_aligned) {
    const int stride = gridDim.x * blockDim.x;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (block_any != nullptr && tid < RENDER_MASK_WORDS * 32) {
        const uint4* cells = reinterpret_cast<const uint4*>(bitfield) + 4 * (size_t)tid;
        const uint4 a = cells[0], b = cells[1], c = cells[2], d = cells[3];
        const bool any = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w | c.x | c.y | c.z | c.w | d.x | d.y | d.z | d.w) != 0u;
        const unsigned long long m = __ballot(any);
        if ((threadIdx.x & 63) == 0) { block_any[tid >> 5] = (uint32_t)m; block_any[(tid >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    if (hits_pairs_aligned) {
        const float2* in2 = reinterpret_cast<const float2*>(hits_in);
        float2* out2 = reinterpret_cast<float2*>(hits);
        for (int i = tid; i < n_rays; i += stride) out2[i] = in2[i];
    } else {
        for (int i = tid; i < 2 * n_rays; i += stride) hits[i] = hits_in[i];
    }
    for (int i = tid; i < n_rays; i += stride) { alive[i] = i; emitted[i] = 0; opacity[i] = 0.f; depth[i] = 0.f; }
    for (int i = tid; i < 3 * n_rays; i += stride) rgb[i] = 0.f;
    int32_t* pl = reinterpret_cast<int32_t*>(plan);
    for (int i = tid; i < (RENDER_MAX_ITERS + 1) * (int)(sizeof(RenderPlan) / 4); i += stride) pl[i] = (i == 0) ? n_rays : 0;
    if (tid == 0) *total = 0ull;
}

// One thread per alive ray, up to N_samples samples each (raymarching.cu:353-403 arithmetic,
// including the calc_dt(..., cascades) quirk via p).  The marching loop only records t in LDS
// (DS traffic does not sit in the vmcnt queue of the dependent bitfield loads); afterwards the
// wave reserves a PACKED range for its samples (one atomic per wave) and every lane expands its
// own samples, so the field is evaluated on real samples only — the reference's dense
// (N_alive, N_samples) buffers carry zero padding for rays that leave early.
// probe_cap == 0: the reference's chunking (a ray marches until it has N samples or leaves, the
// resume point moves only past emitted samples, a ray without samples is retired).
// probe_cap  > 0: at most that many grid probes per ray per iteration (bounds the serial tail of
// rays crossing empty space), the resume point is the next untested lattice point, and a ray is
// retired the moment it reaches its far hit.  Same samples per ray either way.
// lanes of ONE wave exchange data through LDS: DS operations of a wave execute in program order,
// only the compiler has to be kept from reordering them
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool SIMPLE, int NMAX>
__global__ void __launch_bounds__(64)
render_march_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                    float* __restrict__ hits, const int32_t* __restrict__ alive, int32_t* __restrict__ emitted,
                    MarchParams p, RenderPlan* __restrict__ plan, int n_rays, int chunk_scale, int min_samples,
                    int max_samples_total, int probe_cap,
                    float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                    float* __restrict__ ts, int32_t* __restrict__ n_eff, int32_t* __restrict__ offsets,
                    const uint32_t* __restrict__ block_any) {
    __shared__ uint32_t s_any[RENDER_MASK_WORDS];
    __shared__ float s_t[NMAX * 64];        // [sample][lane]; 16 KiB at NMAX = 64 allows two of these waves per SIMD, 8 KiB four
    __shared__ float s_ray[6 * 64];
    __shared__ int s_incl[64];
    // rendering.py:65 `while samples < max_samples`: `samples` grows by N per iteration.  With a probe
    // cap a ray can advance by fewer than N samples per iteration, so that mode budgets the samples
    // per ray instead (emitted[r]); both bound a ray to max_samples (+ at most one chunk).
    const int done = (probe_cap <= 0) ? plan->samples_done : 0;
    const int n_alive = (done < max_samples_total) ? plan->n_alive_raw : 0;
    int N = 0;
    if (n_alive > 0) N = max(min((int)(((long long)chunk_scale * n_rays) / n_alive), NMAX), min_samples);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        plan->n_alive = n_alive; plan->n_step = N;
        plan[1].samples_done = done + N;
    }
    if (blockIdx.x * 64 >= n_alive) return;
    const int lane = threadIdx.x;
    const int n = blockIdx.x * 64 + lane;
    const bool active = n < n_alive;
#ifdef NGP_RENDER_TIMING
    const unsigned long long clk0 = wall_clock64();
    int n_probes = 0, n_short = 0, n_long = 0;           // hops of at most / more than 8 lattice steps (a cell's diagonal is 8 steps long)
#endif
    // rays leaving the object walk the rest of the box cell by cell and emit nothing: a chain of dependent bitfield loads per lane,
    // as long as the longest walk in the wave.  With the 8^3-block bits in LDS the walk through empty blocks needs no global load.
    const uint32_t* any = nullptr;
    if (SIMPLE && block_any != nullptr) {
        s_any[lane] = block_any[lane]; s_any[64 + lane] = block_any[64 + lane];
        wave_lds_fence();
        any = s_any;
    }
    int s = 0, flags = 0;
    Ray ray = {};
    size_t r = 0;
    if (active) {
        r = (size_t)alive[n];
        ray = load_ray(rays_o, rays_d, r);
        float t = hits[2 * r];
        const float t2 = hits[2 * r + 1];
        // march_probe's `hop_slack`: 4 x 2 x the rounding of a crossing time anywhere on this ray (t <= t2, tau - t < 0.25); a ray with a
        // zero direction component has an infinite one and walks cell by cell
        const float hop_slack = 8.0f * (fabsf(t2) * 1.2e-7f + 1.2e-7f + 1.2e-7f * fmaxf(fabsf(ray.ix), fmaxf(fabsf(ray.iy), fabsf(ray.iz))));
        if (probe_cap <= 0) {
            float t_resume = t;
            int iters = 0;
            while (t < t2 && s < N) {
                if (++iters > MARCH_ITER_CAP) { atomicAdd(&g_march_guard[2], 1u); t_resume = t2; break; }
#ifdef NGP_RENDER_TIMING
                ++n_probes;
#endif
                float x, y, z, dt, t_next;
                if (march_probe<SIMPLE>(ray, p, t, x, y, z, dt, t_next, nullptr, any, hop_slack)) {
                    s_t[s * 64 + lane] = t;
                    t += dt; ++s;
                    t_resume = t;
                } else {
#ifdef NGP_RENDER_TIMING
                    if (t_next - t > 8.5f * p.dt_lo) ++n_long; else ++n_short;
#endif
                    t = t_next;
                }
            }
            if (s > 0) hits[2 * r] = t_resume;
            else flags = RENDER_RETIRE;                    // N_eff == 0 (volumerendering.cu:222)
        } else {
            int probes = 0;
            for (;;) {
                if (!(t < t2)) { flags = RENDER_RETIRE; break; }
                if (s >= N || probes >= probe_cap) break;
                ++probes;
                float x, y, z, dt, t_next;
                if (march_probe<SIMPLE>(ray, p, t, x, y, z, dt, t_next, nullptr, any, hop_slack)) {
                    s_t[s * 64 + lane] = t;
                    t += dt; ++s;
                } else {
                    t = t_next;
                }
            }
            hits[2 * r] = t;
            const int e = emitted[r] + s;
            emitted[r] = e;
            if (e >= max_samples_total) flags = RENDER_RETIRE;
        }
    }
    // packed placement: exclusive wave scan of s, one atomic per wave on the iteration's counter
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(&plan->m, total);
    base = __shfl(base, 63, 64);
    if (active) { offsets[n] = base + incl - s; n_eff[n] = s | flags; }
    // cooperative expansion: packed position j of the wave's range -> (ray, k) by a search in the
    // inclusive counts; consecutive lanes write consecutive samples (coalesced streams) instead
    // of every lane scattering its own 8 words per sample
    s_incl[lane] = incl;
    s_ray[lane] = ray.ox; s_ray[64 + lane] = ray.oy; s_ray[128 + lane] = ray.oz;
    s_ray[192 + lane] = ray.dx; s_ray[256 + lane] = ray.dy; s_ray[320 + lane] = ray.dz;
    wave_lds_fence();
    for (int j = lane; j < total; j += 64) {
        int lo = 0;                                   // first ray with incl > j
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (s_incl[lo + step - 1] <= j) lo += step;
        const int k = j - (lo ? s_incl[lo - 1] : 0);
        const float t = s_t[k * 64 + lo];
        const float ox = s_ray[lo], oy = s_ray[64 + lo], oz = s_ray[128 + lo];
        const float dx = s_ray[192 + lo], dy = s_ray[256 + lo], dz = s_ray[320 + lo];
        const size_t o = (size_t)base + j;
        xyzs[3 * o] = fmaf(t, dx, ox); xyzs[3 * o + 1] = fmaf(t, dy, oy); xyzs[3 * o + 2] = fmaf(t, dz, oz);
        dirs[3 * o] = dx; dirs[3 * o + 1] = dy; dirs[3 * o + 2] = dz;
        ts[o] = t; deltas[o] = SIMPLE ? p.dt_lo : calc_dt(t, p);
    }
#ifdef NGP_RENDER_TIMING
    {
        int mx = n_probes, sm = n_probes;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o, 64)); sm += __shfl_xor(sm, o, 64); }
        const int first = (int)__builtin_ctzll(__ballot(n_probes == mx));          // the wave's longest lane and its own mix of hops
        const int l_short = __shfl(n_short, first, 64), l_long = __shfl(n_long, first, 64), l_s = __shfl(s, first, 64);
        if (lane == 0) {
            const unsigned int row = atomicAdd(&g_render_timing_n, 1u);
            if (row < RENDER_TIMING_ROWS) {
                unsigned long long* q = g_render_timing + 6 * (size_t)row;
                q[0] = clk0; q[1] = wall_clock64();
                q[2] = (unsigned long long)(unsigned)mx | ((unsigned long long)(unsigned)sm << 32);
                q[3] = (unsigned long long)(unsigned)N | ((unsigned long long)(unsigned)n_alive << 32);
                q[4] = (unsigned long long)(unsigned)l_short | ((unsigned long long)(unsigned)l_long << 16) | ((unsigned long long)(unsigned)l_s << 32);
                q[5] = blockIdx.x;
            }
        }
    }
#endif
}

// composite_test_fw (volumerendering.cu:219-248) for one iteration, fused with the alive-ray
// compaction (rendering.py:105), the N_eff sum (rendering.py:88) and the hand-over of the
// survivor count to the next iteration's plan and to the host (pinned, lag-polled).
constexpr int RC_THREADS = 1024, RC_WAVES = RC_THREADS / 64, RC_BATCH = 8;
struct __attribute__((packed, aligned(4))) Vec4 { float v[4]; };
__device__ __forceinline__ Vec4 ldv4(const float* p) { return *reinterpret_cast<const Vec4*>(p); }
__global__ void __launch_bounds__(RC_THREADS)
render_composite_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                        const float* __restrict__ deltas, const float* __restrict__ ts,
                        const int32_t* __restrict__ alive_in, int32_t* __restrict__ alive_out,
                        const int32_t* __restrict__ n_eff, const int32_t* __restrict__ offsets, float T_threshold,
                        RenderPlan* __restrict__ plan, float* __restrict__ opacity,
                        float* __restrict__ depth, float* __restrict__ rgb,
                        unsigned long long* __restrict__ total, int32_t* __restrict__ host_count) {
    __shared__ int s_keep[RC_WAVES], s_cnt[RC_WAVES], s_base;
    const int n_alive = plan->n_alive;
    if (n_alive == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { *host_count = 0; __threadfence_system(); }
        return;
    }
    if (blockIdx.x * RC_THREADS >= n_alive) return;
    const int n = blockIdx.x * RC_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool keep = false;
    int cnt = 0, r = 0;
    if (n < n_alive) {
        const int e = n_eff[n];
        cnt = e & 0xffff;
        bool retire = (e & RENDER_RETIRE) != 0;
        r = alive_in[n];
        if (cnt > 0) {
            const size_t base = (size_t)offsets[n];
            float O = opacity[r], D = depth[r], R = rgb[3 * (size_t)r], G = rgb[3 * (size_t)r + 1], B = rgb[3 * (size_t)r + 2];
            float T = 1 - O;
            // the per-sample arithmetic is sequential (volumerendering.cu:229-246); the loads are not:
            // fetch RC_BATCH samples ahead so the chain does not pay one memory latency per sample
            bool stop = false;
            for (int s0 = 0; s0 < cnt && !stop; s0 += RC_BATCH) {
                // 16-byte gathers at 4-byte alignment (legal for global memory on gfx9): 12 instead of 48 load
                // instructions per 8 samples.  Reads past cnt stay inside the workspace (layout slack) and are unused.
                float sg[RC_BATCH], dl[RC_BATCH], tt[RC_BATCH], cr[RC_BATCH], cg[RC_BATCH], cb[RC_BATCH];
                {
                    const size_t o = base + s0;
                    const Vec4 a0 = ldv4(sigmas + o), a1 = ldv4(sigmas + o + 4);
                    const Vec4 b0 = ldv4(deltas + o), b1 = ldv4(deltas + o + 4);
                    const Vec4 c0 = ldv4(ts + o), c1 = ldv4(ts + o + 4);
                    Vec4 q[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) q[k] = ldv4(rgbs + 3 * o + 4 * k);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sg[k] = a0.v[k]; sg[4 + k] = a1.v[k]; dl[k] = b0.v[k]; dl[4 + k] = b1.v[k];
                        tt[k] = c0.v[k]; tt[4 + k] = c1.v[k];
                    }
#pragma unroll
                    for (int k = 0; k < RC_BATCH; ++k) {
                        cr[k] = q[(3 * k) / 4].v[(3 * k) % 4];
                        cg[k] = q[(3 * k + 1) / 4].v[(3 * k + 1) % 4];
                        cb[k] = q[(3 * k + 2) / 4].v[(3 * k + 2) % 4];
                    }
                }
#pragma unroll
                for (int k = 0; k < RC_BATCH; ++k) {
                    if (s0 + k < cnt && !stop) {
                        const float a = 1.0f - __expf(-sg[k] * dl[k]);
                        const float w = a * T;
                        R += w * cr[k]; G += w * cg[k]; B += w * cb[k];
                        D += w * tt[k];
                        O += w;
                        T *= 1.0f - a;
                        if (T <= T_threshold) stop = true;
                    }
                }
            }
            retire = retire || stop;
            opacity[r] = O; depth[r] = D; rgb[3 * (size_t)r] = R; rgb[3 * (size_t)r + 1] = G; rgb[3 * (size_t)r + 2] = B;
        }
        keep = !retire;
    }
    const unsigned long long m = __ballot(keep);
    int c = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) { s_keep[wave] = __popcll(m); s_cnt[wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = 0, q = 0;
        for (int w = 0; w < RC_WAVES; ++w) { k += s_keep[w]; q += s_cnt[w]; }
        s_base = k ? atomicAdd(&plan[1].n_alive_raw, k) : 0;
        if (q) atomicAdd(total, (unsigned long long)q);
    }
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_keep[w];
    if (keep) alive_out[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
    // last workgroup publishes the survivor count to the host
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(&plan->blocks_done, 1);
        if (ticket == (n_alive + RC_THREADS - 1) / RC_THREADS - 1) {
            *host_count = atomicAdd(&plan[1].n_alive_raw, 0);
            __threadfence_system();
        }
    }
}

__global__ void __launch_bounds__(256)
render_finish_kernel(const float* __restrict__ opacity, float* __restrict__ rgb, int n_rays,
                     float bg_r, float bg_g, float bg_b, int blend,
                     const unsigned long long* __restrict__ total, int64_t* __restrict__ total_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && total_out != nullptr) *total_out = (int64_t)*total;
    if (i >= n_rays || !blend) return;
    const float rest = 1 - opacity[i];
    rgb[3 * (size_t)i] = rgb[3 * (size_t)i] + bg_r * rest;
    rgb[3 * (size_t)i + 1] = rgb[3 * (size_t)i + 1] + bg_g * rest;
    rgb[3 * (size_t)i + 2] = rgb[3 * (size_t)i + 2] + bg_b * rest;
}

// The same iteration with ONE WAVE PER RAY, for the late iterations of a frame: a few thousand rays x up to 64 samples each, where a
// thread per ray is one long chain of dependent probes per lane (a launch over 64 rays took 20 us, one over 1 856 rays 51 us).  The
// wave walks the ray's lattice a tile of 64 candidates at a time exactly as march_train_count_wave_kernel does (all candidates probed
// at once with march_probe's arithmetic, skip chains closed by pointer doubling, the orbit of the entry lane on the scalar unit) and
// stops after the ray's N-th sample like the serial loop: same samples, same resume point (t of the last sample + its step), same
// N_eff / retire flags.  16 rays per workgroup: their counts are summed in LDS and the packed range is reserved with one atomic.
// Reference chunking only (probe_cap == 0: the capped mode counts probes, which a tile does not have).
constexpr int RENDER_WAVE_RAYS_PER_WG = 16;
template <bool SIMPLE>
__global__ void __launch_bounds__(64 * RENDER_WAVE_RAYS_PER_WG)
render_march_wave_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                         float* __restrict__ hits, const int32_t* __restrict__ alive,
                         MarchParams p, RenderPlan* __restrict__ plan, int n_rays, int chunk_scale, int min_samples, int max_samples_total,
                         float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                         float* __restrict__ ts, int32_t* __restrict__ n_eff, int32_t* __restrict__ offsets) {
    __shared__ float s_t[RENDER_WAVE_RAYS_PER_WG][64];
    __shared__ int s_cnt[RENDER_WAVE_RAYS_PER_WG];
    __shared__ int s_base;
    const int done = plan->samples_done;
    const int n_alive = (done < max_samples_total) ? plan->n_alive_raw : 0;
    int N = 0;
    if (n_alive > 0) N = max(min((int)(((long long)chunk_scale * n_rays) / n_alive), 64), min_samples);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        plan->n_alive = n_alive; plan->n_step = N;
        plan[1].samples_done = done + N;
    }
    if ((int)blockIdx.x * RENDER_WAVE_RAYS_PER_WG >= n_alive) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int n = (int)blockIdx.x * RENDER_WAVE_RAYS_PER_WG + w;
    const bool active = n < n_alive;
    int s = 0, flags = 0;
    Ray ray = {};
    size_t r = 0;
    if (active) {
        r = (size_t)alive[n];
        ray = load_ray(rays_o, rays_d, r);
        const float t2 = hits[2 * r + 1];
        float t_start = hits[2 * r];
        float pending = -1.0f;                              // landing value of a skip that left the previous tile (< 0: none)
        int tiles = 0;
        for (;;) {
            if (++tiles > MARCH_TILE_CAP) { if (lane == 0) atomicAdd(&g_march_guard[1], 1u); break; }
            // 1. the tile's elements
            float mine = t_start, t_end = t_start;
            if (!(SIMPLE && lattice_tile_const_dt(t_start, p.dt_lo, lane, mine, t_end))) {
                float tt = t_start;
                mine = t_start;
#pragma unroll 8
                for (int j = 0; j < 64; ++j) {
                    mine = (lane == j) ? tt : mine;
                    tt += SIMPLE ? p.dt_lo : calc_dt(tt, p);
                }
                t_end = tt;
            }
            const int nvalid = __popcll(__ballot(mine < t2));                  // the sequence increases: valid lanes are a prefix
            const int entry = pending >= 0 ? __popcll(__ballot(mine < pending)) : 0;
            if (entry >= 64) {
                if (nvalid < 64) break;
                t_start = t_end;
                continue;
            }
            // 2. all candidates at once
            float x, y, z, dt, t_next = 0.f;
            int k = 1;
            const bool occ = march_probe<SIMPLE>(ray, p, mine, x, y, z, dt, t_next, &k);
            const unsigned long long empty_mask = ~__ballot(occ);
            // 3a. where does the chain of skips that starts at an empty lane end (pointer doubling, see march_train_count_wave_kernel)
            int chain = occ ? ((lane << 8) | 64) : ((lane << 8) | (lane + k > 64 ? 64 : lane + k));
#pragma unroll 1
            for (int round = 0; round < 6; ++round) {
                const int h = chain & 0xff;
                const bool go = h < 64 && ((empty_mask >> h) & 1ull);
                const int via = __builtin_amdgcn_ds_bpermute((h & 63) << 2, chain);
                if (go) chain = via;
                if (!__ballot(go)) break;
            }
            // 3b. the orbit of `entry`
            unsigned long long emit = 0ull;
            float next_pending = -1.0f;
            bool finished = false;
            int v = entry;
            while (v < 64) {
                if (v >= nvalid) { finished = true; break; }
                if ((empty_mask >> v) & 1ull) {
                    const int pk = __builtin_amdgcn_readlane(chain, v);
                    const int h = pk & 0xff;
                    if (h >= 64) {
                        next_pending = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_next), pk >> 8));
                        v = 64;
                        break;
                    }
                    v = h;
                    continue;
                }
                const unsigned long long un = empty_mask >> v;
                const int u = un ? v + (int)__builtin_ctzll(un) : 64;
                const int hi = u < nvalid ? u : nvalid;
                if (hi > v) {
                    const int len = hi - v;
                    emit |= (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << v;
                }
                if (nvalid < 64 && u >= nvalid) { finished = true; break; }
                v = u;
            }
            // 4. the ray's next samples, N in all
            const int room = N - s;
            const bool my = (emit >> lane) & 1ull;
            const int rank = __popcll(emit & ((1ull << lane) - 1ull));
            if (my && rank < room) s_t[w][s + rank] = mine;
            const int cnt = __popcll(emit);
            s += cnt < room ? cnt : room;
            if (finished || s >= N) break;
            pending = next_pending;
            if (pending >= 0 && !(pending < t2)) break;
            t_start = t_end;
        }
        wave_lds_fence();
        if (s > 0) {
            const float t_last = s_t[w][s - 1];
            if (lane == 0) hits[2 * r] = t_last + (SIMPLE ? p.dt_lo : calc_dt(t_last, p));      // the serial loop's `t += dt` behind its last sample
        } else {
            flags = RENDER_RETIRE;                           // N_eff == 0 (volumerendering.cu:222)
        }
    }
    if (lane == 0) s_cnt[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int j = 0; j < RENDER_WAVE_RAYS_PER_WG; ++j) total += s_cnt[j];
        s_base = total > 0 ? atomicAdd(&plan->m, total) : 0;
    }
    __syncthreads();
    if (!active) return;
    int base = s_base;
    for (int j = 0; j < w; ++j) base += s_cnt[j];
    if (lane == 0) { offsets[n] = base; n_eff[n] = s | flags; }
    for (int j = lane; j < s; j += 64) {
        const float t = s_t[w][j];
        const size_t o = (size_t)base + j;
        xyzs[3 * o] = fmaf(t, ray.dx, ray.ox); xyzs[3 * o + 1] = fmaf(t, ray.dy, ray.oy); xyzs[3 * o + 2] = fmaf(t, ray.dz, ray.oz);
        dirs[3 * o] = ray.dx; dirs[3 * o + 1] = ray.dy; dirs[3 * o + 2] = ray.dz;
        ts[o] = t; deltas[o] = SIMPLE ? p.dt_lo : calc_dt(t, p);
    }
}

struct RenderLayout {
    size_t hits, alive0, alive1, n_eff, offsets, emitted, plan, total, block_any, xyzs, dirs, deltas, ts, feats, sigmas, rgbs, bytes;
    long long m_cap;
};
RenderLayout render_layout(int n_rays, int chunk_scale, float esf) {
    RenderLayout L;
    const int min_samples = (esf == 0.0f) ? 1 : 4;
    const long long R = n_rays;
    L.m_cap = R * (chunk_scale > min_samples ? chunk_scale : min_samples);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.hits = take(R * 8); L.alive0 = take(R * 4); L.alive1 = take(R * 4); L.n_eff = take(R * 4); L.offsets = take(R * 4); L.emitted = take(R * 4);
    L.plan = take((RENDER_MAX_ITERS + 1) * sizeof(RenderPlan)); L.total = take(8); L.block_any = take(RENDER_MASK_WORDS * 4);
    const long long slack = 64;            // the composite reads whole 8-sample batches
    L.xyzs = take(L.m_cap * 12); L.dirs = take(L.m_cap * 12); L.deltas = take((L.m_cap + slack) * 4); L.ts = take((L.m_cap + slack) * 4);
    L.feats = take(L.m_cap * 64); L.sigmas = take((L.m_cap + slack) * 4); L.rgbs = take((L.m_cap + slack) * 12);
    L.bytes = off;
    return L;
}

// pinned survivor counts + events of the lag-polled frame loop, one set per host thread
struct RenderHost {
    int32_t* counts = nullptr;
    hipEvent_t ev[RENDER_RING] = {};
    int device = -1;
    int init() {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev == device) return 0;
        if (device >= 0) {                                   // the calling thread moved to another GPU: events are per device
            for (int i = 0; i < RENDER_RING; ++i) (void)hipEventDestroy(ev[i]);
            (void)hipHostFree(counts);
            device = -1;
        }
        e = hipHostMalloc(reinterpret_cast<void**>(&counts), RENDER_RING * sizeof(int32_t), hipHostMallocDefault);
        if (e != hipSuccess) return (int)e;
        for (int i = 0; i < RENDER_RING; ++i) {
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
        }
        device = dev;
        return 0;
    }
};
thread_local RenderHost g_render_host;
static int g_render_block_hops = 1;                 // ngp_debug_render_block_hops
static int g_render_wave_rays = 80000;              // ngp_debug_render_wave_rays: iterations with at most this many rays run one wave per ray


double render_wait_limit_s() {
    static const double v = [] { const char* e = getenv("NGP_SPIN_TIMEOUT_S"); const double x = e ? atof(e) : 30.0; return x > 0 ? x : 30.0; }();
    return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {
#pragma GCC visibility push(default)

int ngp_abi_version(void) { return 6; }

int ngp_march_guard_first(float* probe12) {
    NGP_CHECK_PTR(probe12);
    return (int)hipMemcpyFromSymbol(probe12, HIP_SYMBOL(g_march_guard_first), 12 * sizeof(float), 0, hipMemcpyDeviceToHost);
}

int ngp_march_guard_read(uint32_t* counts4, int reset) {
    NGP_CHECK_PTR(counts4);
    hipError_t e = hipMemcpyFromSymbol(counts4, HIP_SYMBOL(g_march_guard), 4 * sizeof(uint32_t), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (reset) {
        const uint32_t z[4] = {0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_march_guard), z, sizeof(z), 0, hipMemcpyHostToDevice);
    }
    return (int)e;
}
const char* ngp_build_arch(void) { return "gfx950"; }

int ngp_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers,
                           const float* half_sizes, int n_rays, int n_voxels, int max_hits,
                           int32_t* hit_cnt, float* hits_t, int64_t* hits_voxel_idx, ngp_stream_t stream) {
    if (n_rays < 0 || n_voxels < 0 || max_hits < 1) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(hit_cnt); NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(hits_voxel_idx);
    if (n_voxels > 0) { NGP_CHECK_PTR(centers); NGP_CHECK_PTR(half_sizes); }
    hipLaunchKernelGGL(ray_prim_intersect_kernel<false>, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, centers, half_sizes, n_rays, n_voxels, max_hits, hit_cnt, hits_t, hits_voxel_idx);
    return NGP_LAUNCH_RESULT();
}

int ngp_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers,
                             const float* radii, int n_rays, int n_spheres, int max_hits,
                             int32_t* hit_cnt, float* hits_t, int64_t* hits_sphere_idx, ngp_stream_t stream) {
    if (n_rays < 0 || n_spheres < 0 || max_hits < 1) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(hit_cnt); NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(hits_sphere_idx);
    if (n_spheres > 0) { NGP_CHECK_PTR(centers); NGP_CHECK_PTR(radii); }
    hipLaunchKernelGGL(ray_prim_intersect_kernel<true>, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, centers, radii, n_rays, n_spheres, max_hits, hit_cnt, hits_t, hits_sphere_idx);
    return NGP_LAUNCH_RESULT();
}

int ngp_ray_aabb_near(const float* rays_o, const float* rays_d, const float* center,
                      const float* half_size, float near_distance, int n_rays, float* hits_t,
                      ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(center); NGP_CHECK_PTR(half_size); NGP_CHECK_PTR(hits_t);
    hipLaunchKernelGGL(ray_aabb_near_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, center, half_size, near_distance, n_rays, hits_t, (float*)nullptr, 0u, 0u);
    return NGP_LAUNCH_RESULT();
}

int ngp_ray_aabb_near_noise(const float* rays_o, const float* rays_d, const float* center,
                            const float* half_size, float near_distance, int n_rays, uint64_t seed,
                            float* hits_t, float* noise, ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(center); NGP_CHECK_PTR(half_size); NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(noise);
    hipLaunchKernelGGL(ray_aabb_near_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, center, half_size, near_distance, n_rays, hits_t, noise, (uint32_t)seed, (uint32_t)(seed >> 32));
    return NGP_LAUNCH_RESULT();
}

int ngp_morton3D(const int32_t* coords, int n, int32_t* indices, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(coords); NGP_CHECK_PTR(indices);
    hipLaunchKernelGGL(morton3D_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, ngp_stream(stream), coords, n, indices);
    return NGP_LAUNCH_RESULT();
}

int ngp_morton3D_invert(const int32_t* indices, int n, int32_t* coords, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(coords); NGP_CHECK_PTR(indices);
    hipLaunchKernelGGL(morton3D_invert_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, ngp_stream(stream), indices, n, coords);
    return NGP_LAUNCH_RESULT();
}

int ngp_packbits(const void* density_grid, int grid_is_half, int n_bytes, float density_threshold,
                 uint8_t* density_bitfield, ngp_stream_t stream) {
    if (n_bytes < 0) return NGP_EINVAL;
    if (n_bytes == 0) return 0;
    NGP_CHECK_PTR(density_grid); NGP_CHECK_PTR(density_bitfield);
    const dim3 grid(ngp_div_up(n_bytes, 256)), block(256);
    if (grid_is_half)
        hipLaunchKernelGGL(packbits_kernel<_Float16>, grid, block, 0, ngp_stream(stream),
                           (const _Float16*)density_grid, n_bytes, density_threshold, (const float*)nullptr, density_bitfield);
    else
        hipLaunchKernelGGL(packbits_kernel<float>, grid, block, 0, ngp_stream(stream),
                           (const float*)density_grid, n_bytes, density_threshold, (const float*)nullptr, density_bitfield);
    return NGP_LAUNCH_RESULT();
}

int ngp_packbits_auto(const float* density_grid, int n_bytes, const float* stats, float density_threshold,
                      uint8_t* density_bitfield, ngp_stream_t stream) {
    if (n_bytes < 0) return NGP_EINVAL;
    if (n_bytes == 0) return 0;
    NGP_CHECK_PTR(density_grid); NGP_CHECK_PTR(density_bitfield); NGP_CHECK_PTR(stats);
    hipLaunchKernelGGL(packbits_kernel<float>, dim3(ngp_div_up(n_bytes, 256)), dim3(256), 0, ngp_stream(stream),
                       density_grid, n_bytes, density_threshold, stats, density_bitfield);
    return NGP_LAUNCH_RESULT();
}

int ngp_density_grid_update(float* density_grid, const float* density_grid_tmp, const float* decay_grid,
                            float decay, int n_cells, float* stats, ngp_stream_t stream) {
    if (n_cells < 0) return NGP_EINVAL;
    if (n_cells == 0) return 0;
    NGP_CHECK_PTR(density_grid); NGP_CHECK_PTR(density_grid_tmp); NGP_CHECK_PTR(stats);
    if ((reinterpret_cast<uintptr_t>(density_grid) | reinterpret_cast<uintptr_t>(density_grid_tmp) | reinterpret_cast<uintptr_t>(decay_grid)) & 15)
        return NGP_EINVAL;                                   // 16-byte accesses
    const int blocks = min(ngp_div_up(ngp_div_up(n_cells, 4), 256), GRID_UPDATE_BLOCKS);
    hipLaunchKernelGGL(density_grid_update_kernel, dim3(blocks), dim3(256), 0, ngp_stream(stream),
                       density_grid, density_grid_tmp, decay_grid, decay, n_cells, stats);
    return NGP_LAUNCH_RESULT();
}

int ngp_raymarching_train_count(const float* rays_o, const float* rays_d, const float* hits_t,
                                const uint8_t* density_bitfield, int cascades, float scale,
                                float exp_step_factor, const float* noise, int grid_size,
                                int max_samples, int n_rays, int64_t* rays_a, int32_t* counter,
                                float* t_scratch, ngp_stream_t stream) {
    return ngp_raymarching_train_count_k(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size, max_samples,
                                         n_rays, rays_a, counter, t_scratch, 0, nullptr, stream);
}

int ngp_raymarching_train_count_k(const float* rays_o, const float* rays_d, const float* hits_t,
                                  const uint8_t* density_bitfield, int cascades, float scale,
                                  float exp_step_factor, const float* noise, int grid_size,
                                  int max_samples, int n_rays, int64_t* rays_a, int32_t* counter,
                                  float* t_scratch, int first_k, int32_t* offs_k, ngp_stream_t stream) {
    if (n_rays < 0 || cascades < 1 || grid_size < 1 || grid_size > 1024 || max_samples < 1) return NGP_EINVAL;
    if (offs_k != nullptr && (first_k < 1 || first_k > 64)) return NGP_EINVAL;
    NGP_CHECK_PTR(counter);
    if (n_rays > 0) {
        NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(density_bitfield);
        NGP_CHECK_PTR(noise); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(t_scratch);
        const MarchParams p = make_march_params(density_bitfield, cascades, grid_size, scale, scale, exp_step_factor, max_samples);
        // Pass 1 runs one WAVE per ray (march_train_count_wave_kernel: 64 candidates of the ray's fixed t-sequence probed per pass,
        // bit-identical to the serial loop, 101 us instead of 280-330 us per 8192-ray batch: profiles/archive_r01_r04/r01_v20_wave_march_kernel_trace.txt,
        // gpurun sweep of round 2: the step time is the same for every placement, the marching stream is busy a third as long).
        const dim3 grid(ngp_div_up((long long)n_rays * 64, 256));
        if (p.simple)
            hipLaunchKernelGGL((march_train_count_wave_kernel<true, false>), grid, dim3(256), 0, ngp_stream(stream),
                               rays_o, rays_d, hits_t, noise, p, max_samples, n_rays, rays_a, t_scratch, (int32_t*)nullptr, MarchPrologue{});
        else
            hipLaunchKernelGGL((march_train_count_wave_kernel<false, false>), grid, dim3(256), 0, ngp_stream(stream),
                               rays_o, rays_d, hits_t, noise, p, max_samples, n_rays, rays_a, t_scratch, (int32_t*)nullptr, MarchPrologue{});
    }
    hipLaunchKernelGGL(march_train_scan_kernel, dim3(1), dim3(1024), 0, ngp_stream(stream), rays_a, n_rays, counter, first_k, offs_k);
    return NGP_LAUNCH_RESULT();
}

int ngp_raymarching_train_write(const float* rays_o, const float* rays_d, const int64_t* rays_a,
                                const float* t_scratch, float scale, float exp_step_factor,
                                int grid_size, int max_samples, int n_rays,
                                float* xyzs, float* dirs, float* deltas, float* ts, ngp_stream_t stream) {
    if (n_rays < 0 || grid_size < 1 || max_samples < 1) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(t_scratch);
    // xyzs..ts may be null only when S == 0, which the kernel never dereferences
    const MarchParams p = make_march_params(nullptr, 1, grid_size, scale, scale, exp_step_factor, max_samples);
    hipLaunchKernelGGL(march_train_write_kernel, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, rays_a, t_scratch, p, max_samples, n_rays, xyzs, dirs, deltas, ts, 0, (int32_t*)nullptr,
                       (int32_t*)nullptr, (const int32_t*)nullptr);
    return NGP_LAUNCH_RESULT();
}

// The native stepper's march: prologue + count in one launch, prefix + expansion in the next (two launches where the API-shaped
// sequence ngp_ray_aabb_near_noise -> ngp_raymarching_train_count [count, scan] -> ngp_raymarching_train_write has four); the same
// arithmetic, the same packing.  counter (pinned host memory, >= 2 x i32) receives {S, R} from the last workgroup of the second
// launch BEFORE it expands its samples: a host that polls it can size the forward's launches while the expansion still runs.
int ngp_march_train_fused(const float* rays_o, const float* rays_d, const float* center, const float* half_size,
                          float near_distance, uint64_t seed, const uint8_t* density_bitfield, int cascades, float scale,
                          float exp_step_factor, int grid_size, int max_samples, int n_rays,
                          float* hits_t, float* noise, int64_t* rays_a, int32_t* counts, int32_t* counter, float* t_scratch,
                          float* xyzs, float* dirs, float* deltas, float* ts, ngp_stream_t stream) {
    if (n_rays < 1 || cascades < 1 || grid_size < 1 || grid_size > 1024 || max_samples < 1) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(center); NGP_CHECK_PTR(half_size); NGP_CHECK_PTR(density_bitfield);
    NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(noise); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(counts); NGP_CHECK_PTR(counter); NGP_CHECK_PTR(t_scratch);
    NGP_CHECK_PTR(xyzs); NGP_CHECK_PTR(dirs); NGP_CHECK_PTR(deltas); NGP_CHECK_PTR(ts);
    if (reinterpret_cast<uintptr_t>(counts) & 15) return NGP_EINVAL;               // (the prefix reads it 16 bytes at a time)
    const MarchParams p = make_march_params(density_bitfield, cascades, grid_size, scale, scale, exp_step_factor, max_samples);
    MarchPrologue pro;
    pro.center = center; pro.half_size = half_size; pro.near_distance = near_distance;
    pro.seed_lo = (uint32_t)seed; pro.seed_hi = (uint32_t)(seed >> 32); pro.hits_out = hits_t; pro.noise_out = noise;
    const dim3 grid(ngp_div_up((long long)n_rays * 64, 256));
    if (p.simple)
        hipLaunchKernelGGL((march_train_count_wave_kernel<true, true>), grid, dim3(256), 0, ngp_stream(stream),
                           rays_o, rays_d, (const float*)nullptr, (const float*)nullptr, p, max_samples, n_rays, rays_a, t_scratch, counts, pro);
    else
        hipLaunchKernelGGL((march_train_count_wave_kernel<false, true>), grid, dim3(256), 0, ngp_stream(stream),
                           rays_o, rays_d, (const float*)nullptr, (const float*)nullptr, p, max_samples, n_rays, rays_a, t_scratch, counts, pro);
    const MarchParams pw = make_march_params(nullptr, 1, grid_size, scale, scale, exp_step_factor, max_samples);
    hipLaunchKernelGGL(march_train_write_kernel, grid, dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, (const int64_t*)rays_a, (const float*)t_scratch, pw, max_samples, n_rays, xyzs, dirs, deltas, ts, 0, (int32_t*)nullptr,
                       (int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)counts, rays_a, counter);
    return NGP_LAUNCH_RESULT();
}

int ngp_raymarching_train_write_k(const float* rays_o, const float* rays_d, const int64_t* rays_a,
                                  const float* t_scratch, float scale, float exp_step_factor,
                                  int grid_size, int max_samples, int n_rays,
                                  float* xyzs, float* dirs, float* deltas, float* ts,
                                  int first_k, int32_t* list_k, int32_t* n_clear, ngp_stream_t stream) {
    if (n_rays < 0 || grid_size < 1 || max_samples < 1 || first_k < 1 || first_k > 64) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(t_scratch); NGP_CHECK_PTR(list_k);
    const MarchParams p = make_march_params(nullptr, 1, grid_size, scale, scale, exp_step_factor, max_samples);
    hipLaunchKernelGGL(march_train_write_kernel, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, rays_a, t_scratch, p, max_samples, n_rays, xyzs, dirs, deltas, ts, first_k, list_k, n_clear,
                       (const int32_t*)nullptr);
    return NGP_LAUNCH_RESULT();
}

int ngp_raymarching_train_write_kc(const float* rays_o, const float* rays_d, const int64_t* rays_a,
                                   const float* t_scratch, float scale, float exp_step_factor,
                                   int grid_size, int max_samples, int n_rays,
                                   float* xyzs, float* dirs, float* deltas, float* ts,
                                   int first_k, const int32_t* offs_k, int32_t* list_k, int32_t* n_clear, ngp_stream_t stream) {
    if (n_rays < 0 || grid_size < 1 || max_samples < 1 || first_k < 1 || first_k > 64) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(t_scratch); NGP_CHECK_PTR(list_k); NGP_CHECK_PTR(offs_k);
    const MarchParams p = make_march_params(nullptr, 1, grid_size, scale, scale, exp_step_factor, max_samples);
    hipLaunchKernelGGL(march_train_write_kernel, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, ngp_stream(stream),
                       rays_o, rays_d, rays_a, t_scratch, p, max_samples, n_rays, xyzs, dirs, deltas, ts, first_k, list_k, n_clear, offs_k);
    return NGP_LAUNCH_RESULT();
}

int ngp_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t,
                         const int64_t* alive_indices, const uint8_t* density_bitfield,
                         int cascades, float scale, float exp_step_factor, int grid_size,
                         int max_samples, int n_samples, int n_alive,
                         float* xyzs, float* dirs, float* deltas, float* ts,
                         int32_t* n_eff_samples, ngp_stream_t stream) {
    if (n_alive < 0 || cascades < 1 || grid_size < 1 || grid_size > 1024 || max_samples < 1 || n_samples < 1) return NGP_EINVAL;
    if (n_alive == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(alive_indices);
    NGP_CHECK_PTR(density_bitfield); NGP_CHECK_PTR(xyzs); NGP_CHECK_PTR(dirs); NGP_CHECK_PTR(deltas);
    NGP_CHECK_PTR(ts); NGP_CHECK_PTR(n_eff_samples);
    // the reference passes `cascades` where calc_dt expects `scale` (raymarching.cu:370,399)
    const MarchParams p = make_march_params(density_bitfield, cascades, grid_size, scale, (float)cascades, exp_step_factor, max_samples);
    // NB: the test kernel's dt clamp uses `cascades` as scale (reference quirk); with esf == 0 dt is dt_lo either way
    if (p.simple)
        hipLaunchKernelGGL(march_test_kernel<true>, dim3(ngp_div_up(n_alive, 64)), dim3(64), 0, ngp_stream(stream),
                           rays_o, rays_d, hits_t, alive_indices, p, n_samples, n_alive, xyzs, dirs, deltas, ts, n_eff_samples);
    else
        hipLaunchKernelGGL(march_test_kernel<false>, dim3(ngp_div_up(n_alive, 64)), dim3(64), 0, ngp_stream(stream),
                           rays_o, rays_d, hits_t, alive_indices, p, n_samples, n_alive, xyzs, dirs, deltas, ts, n_eff_samples);
    return NGP_LAUNCH_RESULT();
}

int ngp_compact_alive(const int64_t* alive_in, const int32_t* n_eff, int n, int64_t* alive_out, int32_t* count,
                      int64_t* total_samples, ngp_stream_t stream) {
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return 0;
    NGP_CHECK_PTR(alive_in); NGP_CHECK_PTR(alive_out); NGP_CHECK_PTR(count);
    hipLaunchKernelGGL(compact_alive_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, ngp_stream(stream),
                       alive_in, n_eff, n, alive_out, count, total_samples);
    return NGP_LAUNCH_RESULT();
}


size_t ngp_render_test_workspace_bytes(int n_rays, int chunk_scale, float exp_step_factor) {
    if (n_rays <= 0 || chunk_scale < 1) return 0;
    return render_layout(n_rays, chunk_scale, exp_step_factor).bytes;
}

#ifdef NGP_RENDER_TIMING
int ngp_debug_render_timing_read(unsigned long long* rows, int max_rows, int reset) {
    unsigned int n = 0;
    hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_render_timing_n), sizeof(n), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return -(int)e;
    if (n > RENDER_TIMING_ROWS) n = RENDER_TIMING_ROWS;
    if ((int)n > max_rows) n = (unsigned)max_rows;
    if (n) e = hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_render_timing), (size_t)n * 48, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return -(int)e;
    if (reset) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_render_timing_n), &z, sizeof(z), 0, hipMemcpyHostToDevice); }
    return (int)n;
}
#endif

int ngp_debug_render_wave_rays(int max_rays) {
    g_render_wave_rays = max_rays < 0 ? 80000 : max_rays;
    return 0;
}

int ngp_debug_render_block_hops(int enabled) {
    g_render_block_hops = enabled ? 1 : 0;
    return 0;
}

int ngp_render_test_frame(const float* rays_o, const float* rays_d, const float* hits_t,
                          const uint8_t* density_bitfield, int cascades, float scale,
                          float exp_step_factor, int grid_size, int max_samples, float T_threshold,
                          const float* xyz_min, const float* xyz_max, const ngp_half* table,
                          const ngp_grid_meta* meta, const ngp_half* density_w, const ngp_half* rgb_w,
                          int n_rays, int chunk_scale, int probe_cap, const float* bg,
                          void* workspace, size_t workspace_bytes,
                          float* opacity, float* depth, float* rgb, int64_t* total_samples,
                          int32_t* n_iterations, ngp_stream_t stream) {
    if (n_iterations) *n_iterations = 0;
    if (n_rays < 0 || chunk_scale < 1 || chunk_scale > 64 || probe_cap < 0 || max_samples <= 0 || !meta) return NGP_EINVAL;
    if (n_rays == 0) return 0;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(hits_t); NGP_CHECK_PTR(density_bitfield);
    NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(table); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(rgb_w);
    NGP_CHECK_PTR(workspace); NGP_CHECK_PTR(opacity); NGP_CHECK_PTR(depth); NGP_CHECK_PTR(rgb);
    const RenderLayout L = render_layout(n_rays, chunk_scale, exp_step_factor);
    if (workspace_bytes < L.bytes) return NGP_EINVAL;
    RenderHost& H = g_render_host;
    int rc = H.init();
    if (rc) return rc;
    char* ws = static_cast<char*>(workspace);
    float* hits = reinterpret_cast<float*>(ws + L.hits);
    int32_t* alive[2] = {reinterpret_cast<int32_t*>(ws + L.alive0), reinterpret_cast<int32_t*>(ws + L.alive1)};
    int32_t* n_eff = reinterpret_cast<int32_t*>(ws + L.n_eff);
    int32_t* emitted = reinterpret_cast<int32_t*>(ws + L.emitted);
    int32_t* offsets = reinterpret_cast<int32_t*>(ws + L.offsets);
    RenderPlan* plan = reinterpret_cast<RenderPlan*>(ws + L.plan);
    unsigned long long* total = reinterpret_cast<unsigned long long*>(ws + L.total);
    float* xyzs = reinterpret_cast<float*>(ws + L.xyzs); float* dirs = reinterpret_cast<float*>(ws + L.dirs);
    float* deltas = reinterpret_cast<float*>(ws + L.deltas); float* ts = reinterpret_cast<float*>(ws + L.ts);
    ngp_half* feats = reinterpret_cast<ngp_half*>(ws + L.feats);
    float* sigmas = reinterpret_cast<float*>(ws + L.sigmas); float* rgbs = reinterpret_cast<float*>(ws + L.rgbs);
    hipStream_t st = ngp_stream(stream);
    const int min_samples = (exp_step_factor == 0.0f) ? 1 : 4;           // rendering.py:60
    // the reference passes `cascades` where calc_dt expects `scale` (raymarching.cu:370,399)
    const MarchParams p = make_march_params(density_bitfield, cascades, grid_size, scale, (float)cascades, exp_step_factor, max_samples);

    // Synthetic-NeRF setting on the 128^3 grid: the marcher crosses 8^3 blocks without an occupied cell in one hop (march_probe), from
    // 512 bytes of block bits in LDS
    uint32_t* block_any = nullptr;
    if (g_render_block_hops && p.simple && grid_size == 128 && (reinterpret_cast<uintptr_t>(density_bitfield) & 15) == 0) block_any = reinterpret_cast<uint32_t*>(ws + L.block_any);
    hipLaunchKernelGGL(render_begin_kernel, dim3(1024), dim3(256), 0, st, hits_t, n_rays, hits, alive[0], emitted, opacity, depth, rgb, plan, total,
                       density_bitfield, block_any, (reinterpret_cast<uintptr_t>(hits_t) & 7) == 0 ? 1 : 0);
    constexpr int LAG = 2;
    long long bound = n_rays;
    int it = 0;
    for (; it < RENDER_MAX_ITERS; ++it) {
        if (it >= LAG) {
            const int j = (it - LAG) % RENDER_RING;
            // bounded wait: a first quick poll (the event is two iterations old and usually reached), then the blocking wait is
            // replaced by polling with a deadline, so a kernel that never finishes turns into NGP_ETIMEOUT instead of a hung caller
            hipError_t e = hipEventQuery(H.ev[j]);
            if (e == hipErrorNotReady) {
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(render_wait_limit_s());
                int polls = 0;
                while ((e = hipEventQuery(H.ev[j])) == hipErrorNotReady) {
                    if ((++polls & 255) == 0 && std::chrono::steady_clock::now() > t_end) return NGP_ETIMEOUT;
                }
            }
            if (e != hipSuccess) return (int)e;
            bound = H.counts[j];               // survivors after iteration it-LAG >= alive rays now
            if (bound <= 0) break;
        }
        RenderPlan* pl = plan + it;
        const dim3 mgrid(ngp_div_up(bound, 64));
        // the reference's chunking (chunk_scale 1, no probe cap) takes up to 64 samples per ray and iteration (rendering.py:72); the
        // regrouping modes cap a ray's samples per iteration at 32, which halves the marcher's LDS tile and doubles its waves per CU
        const bool regroup = chunk_scale > 1 || probe_cap > 0;
#define NGP_RENDER_MARCH(S, NM) hipLaunchKernelGGL((render_march_kernel<S, NM>), mgrid, dim3(64), 0, st, rays_o, rays_d, hits, alive[it & 1], emitted, p, pl, \
                                                   n_rays, chunk_scale, min_samples, max_samples, probe_cap, xyzs, dirs, deltas, ts, n_eff, offsets, block_any)
        // iteration 0 marches EVERY ray for N = max(chunk_scale, min_samples) samples (n_alive = n_rays): with a tile of 8 samples per
        // ray instead of 32 / 64 the longest walks of the frame run at the SIMDs' full wave count
        const bool first = it == 0 && chunk_scale <= 8;
        // late iterations (few rays, many samples each): one wave per ray (render_march_wave_kernel); `bound` is the survivor count of
        // two iterations ago, an upper bound of the rays alive now.  Crossover measured on the trained fields (tools/frame_wave_ab.py:
        // 0 / 16 384 / 80 000 / 200 000 rays -> 727 / 780 / 824 / 799 FPS on `lego`, 255 / 258 / 263 / 223 on `lego_hard`)
        const bool per_wave = !regroup && it >= LAG && bound <= (long long)g_render_wave_rays;
        if (per_wave) {
            const dim3 wgrid(ngp_div_up(bound, RENDER_WAVE_RAYS_PER_WG)), wblock(64 * RENDER_WAVE_RAYS_PER_WG);
            if (p.simple) hipLaunchKernelGGL((render_march_wave_kernel<true>), wgrid, wblock, 0, st, rays_o, rays_d, hits, alive[it & 1], p, pl, n_rays, chunk_scale,
                                             min_samples, max_samples, xyzs, dirs, deltas, ts, n_eff, offsets);
            else hipLaunchKernelGGL((render_march_wave_kernel<false>), wgrid, wblock, 0, st, rays_o, rays_d, hits, alive[it & 1], p, pl, n_rays, chunk_scale,
                                    min_samples, max_samples, xyzs, dirs, deltas, ts, n_eff, offsets);
        } else if (p.simple) {
            if (first) NGP_RENDER_MARCH(true, 8); else if (regroup) NGP_RENDER_MARCH(true, 32); else NGP_RENDER_MARCH(true, 64);
        } else {
            if (first) NGP_RENDER_MARCH(false, 8); else if (regroup) NGP_RENDER_MARCH(false, 32); else NGP_RENDER_MARCH(false, 64);
        }
#undef NGP_RENDER_MARCH
        const int n_max = regroup ? 32 : 64;
        long long m_bound = (long long)chunk_scale * n_rays;
        if (n_max * bound < m_bound) m_bound = n_max * bound;
        if ((long long)min_samples * bound > m_bound) m_bound = (long long)min_samples * bound;
        if (m_bound > L.m_cap) m_bound = L.m_cap;
        rc = ngp_hashgrid_fwd_n(xyzs, xyz_min, xyz_max, table, meta, (int)m_bound, &pl->m, feats, stream);
        if (rc) return rc;
        rc = ngp_field_fwd_n(feats, dirs, density_w, rgb_w, (int)m_bound, &pl->m, sigmas, rgbs, nullptr, stream);   // h stays in registers
        if (rc) return rc;
        hipLaunchKernelGGL(render_composite_kernel, dim3(ngp_div_up(bound, RC_THREADS)), dim3(RC_THREADS), 0, st, sigmas, rgbs, deltas, ts,
                           alive[it & 1], alive[(it + 1) & 1], n_eff, offsets, T_threshold, pl, opacity, depth, rgb, total,
                           H.counts + it % RENDER_RING);
        hipError_t e = hipEventRecord(H.ev[it % RENDER_RING], st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(render_finish_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, st, opacity, rgb, n_rays,
                       bg ? bg[0] : 0.f, bg ? bg[1] : 0.f, bg ? bg[2] : 0.f, bg ? 1 : 0, total, total_samples);
    if (n_iterations) *n_iterations = it;
    return NGP_LAUNCH_RESULT();
}

}  // extern "C"
