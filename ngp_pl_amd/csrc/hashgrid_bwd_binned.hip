// Hash-grid backward w.r.t. the table, binned variant (reference semantics: tiny-cuda-nn's
// kernel_grid_backward scatter-add as configured at /root/reference/models/networks.py:36-48).
//
// What was measured on MI355X (profiles/archive_r01_r04/r01_hashgrid_bwd_experiments.txt, tools/lds_atomic_bench.hip):
//  * a slice owner of the one-pass kernel (hashgrid_bwd_sliced_kernel) walks ALL samples of its
//    level to find the 1/19 of the corner updates that land in its slice, and it issues 8 LDS
//    float atomics per wave and sample; both cost about the same (VALU ~130 us, LDS ~106 us);
//  * an LDS FLOAT atomic costs ~3 cycles per active lane, an LDS 64-bit INTEGER atomic ~8-12
//    cycles per wave instruction whatever the lane count (16x cheaper on a full wave, and 4x
//    cheaper when all lanes hit the same address);
//  * the 8 corners of a sample fall into only ~4 slices of a level (x and x+1 are neighbours in
//    the table, the four (y,z) combinations scatter).
// Hence:
//  1. a binning pass writes, per (level, slice), the list of samples that touch the slice -- on
//     hashed levels as (sample, corner pair) entries, so that an owner forms 2 corners, not 8;
//  2. slice owners walk only their list and accumulate in 64-bit fixed point (2^-24 units; one
//     update saturates at |w*g| = 128 in loss-scaled units, typical values are 1e-5..1e-1;
//     exact sums, deterministic: integer adds commute -- the f16 atomics of the one-pass kernel
//     and of tiny-cuda-nn round after every add, in arrival order);
//  3. slices are small (16 B per entry: 6912 entries in 108 KiB of LDS) and numerous (~1070 tasks
//     with the dense levels' splits), pulled from a queue by 256 persistent workgroups, the
//     expensive (dense) ones first.
// Levels with few slices (the coarse dense ones: a z-slab can hold most of the scene) additionally split
// their lists over K tasks whose partial tables are summed by a small merge kernel (no atomics anywhere).
#include "hashgrid_common.h"
#include <hip/hip_fp16.h>

using namespace ngp_grid;

namespace {

#ifndef NGP_SLICE2
#define NGP_SLICE2 6912
#endif
#ifndef NGP_APPLY_WGS
#define NGP_APPLY_WGS 256
#endif
constexpr uint32_t SLICE2 = NGP_SLICE2;      // entries per task: 2 x int64 each -> 108 KiB of LDS at 6912
constexpr int MAX_SLICES = ((1 << 19) + SLICE2 - 1) / SLICE2 + 4;   // 2^19 / SLICE2 slices per hashed level (76 at 6912)
constexpr float FIX_SCALE = 16777216.0f;     // 2^24 units per 1.0 (f16 subnormal spacing is 2^-24)
constexpr int BIN_THREADS = 512;             // binning workgroup: 4 of them are resident per CU (the pass is latency-bound)
#ifndef NGP_BIN_SPT
#define NGP_BIN_SPT 2
#endif
constexpr int BIN_SPT = NGP_BIN_SPT;         // samples per thread
constexpr int CHUNK = BIN_THREADS * BIN_SPT; // samples per binning workgroup ("chunk")
constexpr int CHUNK_SLOTS = CHUNK * 8;       // list entries a chunk can produce for one level (<= 8 slices per sample)
constexpr int MAX_CHUNKS = 4608;             // 4.7 M samples: the first steps of a run (every cell still "occupied": ~245 samples per ray, 2 M for 8192
                                             // rays, 4 M for 16 384) stay on this exact, deterministic path instead of the one-pass kernel, whose f16 LDS
                                             // atomics round in arrival order -- round 3's bench runs diverged from step 0 on and landed on operating
                                             // points 15 % apart.  (Two directory rows of this length sit in LDS next to the accumulators: 36 KB.)
// (Payload list entries -- the slice owner handed corner indices, gradient and fixed-point weights in a 12-byte entry, no gathers on its
// side -- were built and measured in round 3: owners 14 % faster, binning twice as slow, 132 -> 145 us for the stage; removed in round 5.
// profiles/archive_r01_r04/r03_table_backward_experiments.txt.)
#ifndef NGP_APPLY_THREADS
#define NGP_APPLY_THREADS 1024
#endif
#ifndef NGP_APPLY_WAVES_PER_EU
#define NGP_APPLY_WAVES_PER_EU 1
#endif
#ifndef NGP_APPLY_B
#define NGP_APPLY_B 11
#endif
#ifndef NGP_APPLY_WRITEOUT_BATCHED
#define NGP_APPLY_WRITEOUT_BATCHED 0          // accumulators per thread read together at write-out; 0: one at a time (round 2; A/B builds)
#endif
#ifndef NGP_DIR_BY_WAVE0
#define NGP_DIR_BY_WAVE0 1                    // 0: every thread fetches its part of the next directory row under the write-out (rounds 4-6; A/B builds)
#endif
#ifndef NGP_DENSE_B
#define NGP_DENSE_B 2                         // dense levels: entries per lane in flight (round 6, same box: 2 -> 122.8 us alone, 4 -> 125.6, 8 -> 128.7)
#endif
constexpr int APPLY_THREADS = NGP_APPLY_THREADS;

struct BinPlan {
    int32_t n_levels;
    int32_t n_slices[NGP_MAX_LEVELS];
    int32_t k_split[NGP_MAX_LEVELS];
    int32_t first_task[NGP_MAX_LEVELS + 1];  // tasks are numbered level by level in `order`
    int32_t order[NGP_MAX_LEVELS];           // levels, most expensive tasks first
    int64_t part_off[NGP_MAX_LEVELS];        // K-split levels: entry offset of the level's K partial tables in ws.partial
    int64_t pool_off[NGP_MAX_LEVELS];        // the level's list slots in ws.pool, in 4-byte words
    int32_t entry_words[NGP_MAX_LEVELS];     // words per list entry: 1 = sample id (dense levels), 3 = payload entry (hashed levels)
    int32_t n_tasks, n_chunks;
};

// Lists are stored chunk-wise: the binning workgroup of (level, chunk) owns the fixed slot
// pool[(level * n_chunks + chunk) * CHUNK_SLOTS ...] and writes its entries there grouped by slice;
// dir[((level * MAX_SLICES + s) * n_chunks + chunk] = (start in the slot << 16) | count of slice s's
// segment (one coalesced row per slice owner).  No atomics, no capacity to exceed, deterministic.
struct BinWs {
    float2* partial;     // partial gradient tables (f32) of the K-split (dense) levels, merged by merge_kernel
    long long* timing;   // NGP_BIN_TIMING builds only: [task][4] timestamps
    int32_t* queue;      // [0] next task
    int32_t* dir;
    int32_t* pool;
};

// ---- pass 1: which slices does a sample touch ------------------------------------------------
__global__ void __launch_bounds__(BIN_THREADS)
bin_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
           const half2_t* __restrict__ dfeats, GridMeta meta, BinPlan plan, BinWs ws, int n_samples,
           const int32_t* __restrict__ active, const int32_t* __restrict__ n_active) {
    __shared__ int s_cnt[MAX_SLICES], s_pre[MAX_SLICES + 1];
    __shared__ int32_t s_stage[CHUNK_SLOTS];
    const int n_chunks = plan.n_chunks;
    const int level = (int)blockIdx.x / n_chunks;
    const int chunk = (int)blockIdx.x - level * n_chunks;
    const int n = n_active ? min(*n_active, n_samples) : n_samples;
    const int ns = plan.n_slices[level];
    if (blockIdx.x == 0 && threadIdx.x < 16) ws.queue[threadIdx.x] = 0;   // the slice owners' task counters, one per launch group (they start after this kernel)
    int32_t* __restrict__ dir = ws.dir + (size_t)level * MAX_SLICES * n_chunks + chunk;     // + s * n_chunks
    if (chunk * CHUNK >= n) return;                                // nothing here, and no slice owner reads this chunk's directory column:
                                                                   // they stop at the last chunk that holds live samples (apply_kernel)
    for (int i = threadIdx.x; i < ns; i += BIN_THREADS) s_cnt[i] = 0;
    __syncthreads();
    const uint32_t res = meta.resolution[level];
    const uint32_t size = meta.offset[level + 1] - meta.offset[level];
    const bool hashed = level_is_hashed(res, size);
    const Box box = load_box(xyz_min, xyz_max);
    int jj[BIN_SPT], sid[BIN_SPT][8], loc[BIN_SPT][8], tag[BIN_SPT][8], n_mine[BIN_SPT];
    half2_t gg[BIN_SPT]; int src[BIN_SPT]; float xin[BIN_SPT][3];
#pragma unroll
    for (int t = 0; t < BIN_SPT; ++t) {               // all loads of the thread's samples before any use
        jj[t] = chunk * CHUNK + t * BIN_THREADS + (int)threadIdx.x;
        const int jc = min(jj[t], n - 1);
        gg[t] = dfeats[(size_t)level * n_samples + jc];
        src[t] = active ? active[jc] : jc;
    }
#pragma unroll
    for (int t = 0; t < BIN_SPT; ++t) { xin[t][0] = x[3 * (size_t)src[t]]; xin[t][1] = x[3 * (size_t)src[t] + 1]; xin[t][2] = x[3 * (size_t)src[t] + 2]; }
#pragma unroll
    for (int t = 0; t < BIN_SPT; ++t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { sid[t][q] = -1; tag[t][q] = 0; }
        n_mine[t] = 0;
        if (jj[t] < n && (gg[t][0] != (_Float16)0 || gg[t][1] != (_Float16)0)) {
            uint32_t p[3], idx[8]; float f[3];
            cell_of_loaded(xin[t], box, meta.scale[level], p, f);
            if (hashed) corner_indices<true>(p, res, size, idx);
            else corner_indices<false>(p, res, size, idx);
            if (hashed) {
                // entry per (sample, corner pair): the corners (x, .) and (x+1, .) are table neighbours and almost
                // always share a slice; the slice owner then forms only that pair (half the index and LDS work of a
                // whole-sample entry, and no in-slice test that fails 3 times out of 4)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int s0 = (int)(idx[2 * k] / SLICE2), s1 = (int)(idx[2 * k + 1] / SLICE2);
                    sid[t][2 * k] = s0; tag[t][2 * k] = k;
                    sid[t][2 * k + 1] = (s1 != s0) ? s1 : -1; tag[t][2 * k + 1] = k;
                }
                n_mine[t] = 8;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int sl = (int)(idx[c] / SLICE2);
                    bool dup = false;
#pragma unroll
                    for (int q = 0; q < c; ++q) dup = dup || (q < n_mine[t] && sid[t][q] == sl);
                    if (!dup) {
                        // compact the distinct slice ids to the front (n_mine is small: 1 or 2 slabs)
#pragma unroll
                        for (int q = 0; q < 8; ++q) if (q == n_mine[t]) sid[t][q] = sl;
                        ++n_mine[t];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < BIN_SPT; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < n_mine[t] && sid[t][q] >= 0) loc[t][q] = atomicAdd(&s_cnt[sid[t][q]], 1);
    __syncthreads();
    if (threadIdx.x < 64) {                       // exclusive prefix of the per-slice counts (ns <= 80): two wave scans
        int run = 0;
        for (int b0 = 0; b0 < ns; b0 += 64) {
            const int i = b0 + (int)threadIdx.x;
            const int v = (i < ns) ? s_cnt[i] : 0;
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += u; }
            if (i < ns) s_pre[i] = run + incl - v;
            run += __shfl(incl, 63, 64);
        }
        if (threadIdx.x == 0) s_pre[ns] = run;
    }
    __syncthreads();
    int32_t* __restrict__ slot = ws.pool + plan.pool_off[level] + (size_t)chunk * CHUNK_SLOTS * plan.entry_words[level];
    {
#pragma unroll
        for (int t = 0; t < BIN_SPT; ++t)
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < n_mine[t] && sid[t][q] >= 0) s_stage[s_pre[sid[t][q]] + loc[t][q]] = (jj[t] << 2) | tag[t][q];
        __syncthreads();
        const int total = s_pre[ns];
        for (int i = threadIdx.x; i < total; i += BIN_THREADS) slot[i] = s_stage[i];
    }
    for (int i = threadIdx.x; i < ns; i += BIN_THREADS) dir[(size_t)i * n_chunks] = (s_pre[i] << 16) | s_cnt[i];
}

// ---- pass 2: slice owners -------------------------------------------------------------------

// v -> round-to-nearest-even(v x 2^24) as a 64-bit integer, any |v| up to the f16 range (x 2^24 = 2^40): through the double
// 1.5 x 2^52 + v x 2^24, whose mantissa IS that integer in two's complement (one exact FMA; the magic's low word is zero, so only the
// high word needs the subtraction).  Rounds 1-5 converted through v_cvt_i32_f32, which saturates at |v| = 128: unreachable at the
// fixed loss scale 128 (feature gradients ~1e-5), a silent clip under the dynamic one, which keeps the largest gradients near the
// top of the f16 range (round 6).  Same integers as before wherever the old conversion did not saturate.
__device__ __forceinline__ long long fix_q64(float v) {
    const double y = __builtin_fma((double)v, (double)FIX_SCALE, 6755399441055744.0);
    return __builtin_bit_cast(long long, y) - 0x4338000000000000LL;
}

__device__ __forceinline__ void lds_add_fixed(long long* acc, float v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)fix_q64(v));   // ds_add_u64
}

// One pass for the lanes of a wave over whole-sample entries (dense levels): all 8 corners, in-slice test each.
template <int B>
__device__ __forceinline__ void apply_entries(long long* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, float scale,
                                              const float* __restrict__ x, const Box& box, const half2_t* __restrict__ g_level,
                                              const int32_t* __restrict__ active, const int (&ee)[B], const bool (&ok)[B]) {
    int src[B]; half2_t g[B]; float px[B][3];
#pragma unroll
    for (int b = 0; b < B; ++b) { const int jj = ee[b] >> 2; g[b] = g_level[jj]; src[b] = active ? active[jj] : jj; }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const float* __restrict__ xp = x + 3 * (size_t)src[b];
        px[b][0] = xp[0]; px[b][1] = xp[1]; px[b][2] = xp[2];
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        if (!ok[b]) continue;
        const float g0 = (float)g[b][0], g1 = (float)g[b][1];
        uint32_t p[3], idx[8]; float f[3];
        cell_of_loaded(px[b], box, scale, p, f);
        corner_indices<false>(p, res, size, idx);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t local = idx[c] - lo;
            if (local < len) {
                const float w = corner_weight(c, f);
                lds_add_fixed(lds + 2 * local, w * g0);
                lds_add_fixed(lds + 2 * local + 1, w * g1);
            }
        }
    }
}

// The same for (sample, corner pair) entries of a hashed level: entry = (j << 2) | (cy + 2 cz); only the pair
// (x, y+cy, z+cz), (x+1, y+cy, z+cz) is formed -- index arithmetic and weight order of corner_indices()/corner_weight().
template <int B>
__device__ __forceinline__ void apply_pairs(long long* lds, uint32_t lo, uint32_t len, uint32_t size, float scale,
                                            const float* __restrict__ x, const Box& box, const half2_t* __restrict__ g_level,
                                            const int32_t* __restrict__ active, const int (&ee)[B], const bool (&ok)[B]) {
    int src[B]; half2_t g[B]; float px[B][3];
#pragma unroll
    for (int b = 0; b < B; ++b) { const int jj = ee[b] >> 2; g[b] = g_level[jj]; src[b] = active ? active[jj] : jj; }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const float* __restrict__ xp = x + 3 * (size_t)src[b];
        px[b][0] = xp[0]; px[b][1] = xp[1]; px[b][2] = xp[2];
    }
    const uint32_t mask = size - 1u;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        if (!ok[b]) continue;
        const float g0 = (float)g[b][0], g1 = (float)g[b][1];
        uint32_t p[3]; float f[3];
        cell_of_loaded(px[b], box, scale, p, f);
        const uint32_t cy = (uint32_t)ee[b] & 1u, cz = ((uint32_t)ee[b] >> 1) & 1u;
        const uint32_t h = ((p[1] + cy) * PRIME_Y) ^ ((p[2] + cz) * PRIME_Z);
        const uint32_t l0 = ((p[0] ^ h) & mask) - lo, l1 = (((p[0] + 1u) ^ h) & mask) - lo;
        const float wy = cy ? f[1] : 1.f - f[1], wz = cz ? f[2] : 1.f - f[2];
        const float w0 = ((1.f - f[0]) * wy) * wz, w1 = (f[0] * wy) * wz;
        if (l0 < len) { lds_add_fixed(lds + 2 * l0, w0 * g0); lds_add_fixed(lds + 2 * l0 + 1, w0 * g1); }
        if (l1 < len) { lds_add_fixed(lds + 2 * l1, w1 * g0); lds_add_fixed(lds + 2 * l1 + 1, w1 * g1); }
    }
}

// Hashed levels: a wave takes B chunk segments per trip (typically ~57 entries each), lane = entry:
// coalesced entry / gradient / position streams, all of a trip's loads in flight together.
// (Measured and not kept, profiles/archive_r01_r04/r03_table_backward_experiments.txt: lanes of one instruction taken 8 entries apart and from 8
// different chunks, so that runs of consecutive samples in one cell never meet in one ds_add_u64 -- hashed tasks 16.4 -> 23.3 us:
// same-address adds are not what they wait for, the strided entry loads cost more than the conflicts.)
template <int B>
__device__ __forceinline__ void apply_segments_hashed(long long* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, float scale,
                                                      const float* __restrict__ x, const Box& box, const half2_t* __restrict__ g_level,
                                                      const int32_t* __restrict__ active, const int32_t* __restrict__ pool_level,
                                                      const int* s_dir, int n_chunks, int part, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = APPLY_THREADS / 64;
    for (int c0 = part + K * wave; c0 < n_chunks; c0 += K * NW * B) {
        int start[B], cnt[B], maxcnt = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int c = c0 + b * K * NW;
            const int d = (c < n_chunks) ? s_dir[c] : 0;
            start[b] = c * CHUNK_SLOTS + (d >> 16); cnt[b] = d & 0xffff;
            maxcnt = max(maxcnt, cnt[b]);
        }
        for (int off = 0; off < maxcnt; off += 64) {               // one pass unless a segment has more than 64 entries
            int ee[B]; bool ok[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                ok[b] = off + lane < cnt[b];
                ee[b] = ok[b] ? pool_level[(size_t)start[b] + off + lane] : 0;
            }
            apply_pairs<B>(lds, lo, len, size, scale, x, box, g_level, active, ee, ok);
        }
    }
}

// Dense (coarse) levels: a segment holds up to every sample of its chunk, and dozens of consecutive
// samples of a ray sit in the same cell, i.e. consecutive entries hit the same 8 accumulators
// (measured: an LDS add_u64 with 8 lanes per address costs 64 cycles instead of 11.5).  Lane L
// therefore walks the contiguous run [L*R, (L+1)*R) of the segment: the lanes of a wave are a whole
// run apart, on different rays.
#ifndef NGP_DENSE_RUNS
#define NGP_DENSE_RUNS 1                 // 0: every sample's 16 updates go to LDS one by one (round 2; kept for A/B builds)
#endif
template <int B>
__device__ __forceinline__ void apply_segments_dense(long long* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, float scale,
                                                     const float* __restrict__ x, const Box& box, const half2_t* __restrict__ g_level,
                                                     const int32_t* __restrict__ active, const int32_t* __restrict__ pool_level,
                                                     const int* s_dir, int n_chunks, int part, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = APPLY_THREADS / 64;
    for (int c = part + K * wave; c < n_chunks; c += K * NW) {
        const int d = s_dir[c];
        const int cnt = d & 0xffff;
        const size_t start = (size_t)c * CHUNK_SLOTS + (d >> 16);
        const int R = (cnt + 63) >> 6;
        for (int t0 = 0; t0 < R; t0 += B) {
            int jj[B]; bool ok[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int i = lane * R + t0 + b;
                ok[b] = (t0 + b < R) && i < cnt;
                jj[b] = ok[b] ? pool_level[start + i] : 0;
            }
            apply_entries<B>(lds, lo, len, res, size, scale, x, box, g_level, active, jj, ok);
        }
    }
}

// Within a lane's run consecutive entries are consecutive samples of a ray (a chunk's list keeps the order in which the
// binning waves arrived: blocks of 64 consecutive samples), i.e. they sit in the same cell for 10-40 steps on the coarse
// levels: the 8 corner sums are kept in REGISTERS -- as 64-bit fixed point, the accumulators' own arithmetic: exact, so
// neither the run boundaries nor the list order can change a bit of the result -- and reach LDS when the cell changes.
__device__ __forceinline__ void flush_cell(long long* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, uint32_t key,
                                           const long long (&acc)[16]) {
    const uint32_t r2 = res * res;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint32_t i = key + (c & 1) + ((c >> 1) & 1) * res + (c >> 2) * r2;
        i = (i >= size) ? i - size : i;
        const uint32_t local = i - lo;
        if (local < len) {
            if (acc[2 * c] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(lds + 2 * local), (unsigned long long)acc[2 * c]);
            if (acc[2 * c + 1] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(lds + 2 * local + 1), (unsigned long long)acc[2 * c + 1]);
        }
    }
}
template <int B>
__device__ __forceinline__ void apply_segments_dense_runs(long long* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, float scale,
                                                          const float* __restrict__ x, const Box& box, const half2_t* __restrict__ g_level,
                                                          const int32_t* __restrict__ active, const int32_t* __restrict__ pool_level,
                                                          const int* s_dir, int n_chunks, int part, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = APPLY_THREADS / 64;
    const uint32_t top = res - 1u, r2 = res * res;
    for (int c = part + K * wave; c < n_chunks; c += K * NW) {
        const int d = s_dir[c];
        const int cnt = d & 0xffff;
        const size_t start = (size_t)c * CHUNK_SLOTS + (d >> 16);
        const int R = (cnt + 63) >> 6;
        long long acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0;
        uint32_t cur = 0xffffffffu;
        for (int t0 = 0; t0 < R; t0 += B) {
            int jj[B]; bool ok[B]; half2_t g[B]; float px[B][3];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int i = lane * R + t0 + b;
                ok[b] = (t0 + b < R) && i < cnt;
                jj[b] = ok[b] ? (pool_level[start + i] >> 2) : 0;
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                g[b] = g_level[jj[b]];
                const float* __restrict__ xp = x + 3 * (size_t)(active ? active[jj[b]] : jj[b]);
                px[b][0] = xp[0]; px[b][1] = xp[1]; px[b][2] = xp[2];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (!ok[b]) continue;
                const float g0 = (float)g[b][0], g1 = (float)g[b][1];
                uint32_t p[3]; float f[3];
                cell_of_loaded(px[b], box, scale, p, f);
                const uint32_t key = min(p[0], top) + min(p[1], top) * res + min(p[2], top) * r2;       // border clamp: corner_indices<false>
                if (key != cur) {
                    if (cur != 0xffffffffu) flush_cell(lds, lo, len, res, size, cur, acc);
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0;
                    cur = key;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float w = corner_weight(q, f);
                    acc[2 * q] += fix_q64(w * g0);                                                          // lds_add_fixed()'s quantisation
                    acc[2 * q + 1] += fix_q64(w * g1);
                }
            }
        }
        if (cur != 0xffffffffu) flush_cell(lds, lo, len, res, size, cur, acc);
    }
}

// Workgroup barrier for data that lives in LDS only (accumulators, task ids, directory rows).  __syncthreads() carries a workgroup-scope
// fence, and gfx9's single vmcnt makes that fence wait for EVERY outstanding global access of the wave: behind the write-out that is the
// round trip of its gradient stores, which no thread of the launch reads (round 6: write-out phase 2.8 -> 1.6 us per task,
// profiles/r06_table_backward.txt).  The "memory" clobber keeps the compiler from moving LDS accesses across it.
// An entry's sum -> the f16 the gradient table holds, SATURATED: thousands of samples can sum past the f16 range when no single one
// leaves it (those the field backward's overflow guard watches); +-65504 instead of inf keeps the optimizer's moments finite.
__device__ __forceinline__ _Float16 sat_f16(float v) { return (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f); }

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (Applying the hashed levels' Adam update in this kernel's write-out -- three designs, the last with wave specialisation -- was built and
// measured in round 4: the optimizer's stream is not hidden under the owners' latencies, the owners slow down by as much as the stream takes
// alone; removed in round 5.  profiles/archive_r01_r04/r04_step_ab.txt (d), (d2).)
template <bool UNUSED>       // (the template argument only keeps the kernel's name in the round 3-4 traces: apply_kernel<false>)
__global__ void __launch_bounds__(APPLY_THREADS, NGP_APPLY_WAVES_PER_EU)
apply_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
             const half2_t* __restrict__ dfeats, GridMeta meta, BinPlan plan, BinWs ws, int n_samples,
             const int32_t* __restrict__ active, const int32_t* __restrict__ n_active, half2_t* __restrict__ grad_table,
             int group, int task_begin, int task_end) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    long long* lds = reinterpret_cast<long long*>(smem_raw);
    __shared__ int s_task[2];                                          // s_task[k & 1]: id of the workgroup's k-th task
    __shared__ int s_dir2[2][MAX_CHUNKS];                              // directory row of the k-th task in s_dir2[k & 1]
    const Box box = load_box(xyz_min, xyz_max);
    const int dir_pitch = plan.n_chunks;                               // the directory / slot layout is planned for n_samples ...
    // ... the loops stop at the last chunk that holds live samples: late in training a few per cent of the marched samples are
    // live, and a task's fixed cost was dominated by walking hundreds of empty directory entries
    const int n_live = n_active ? min(*n_active, n_samples) : n_samples;
    const int n_chunks = min(plan.n_chunks, (n_live + CHUNK - 1) / CHUNK);
    const int tid = threadIdx.x;
    // TWO barriers per task, none of them behind a global round trip (round 4; three before, the first behind the load of the
    // task's directory row): the id of task k+1 is requested from the queue at the top of task k and published at the barrier
    // behind task k's accumulation; its directory row is then loaded into registers, rides under task k's write-out and is
    // parked in the other half of s_dir2 at the closing barrier.  The accumulators are cleared by the write-out pass that
    // reads them (the first task finds them cleared by the prologue).
    constexpr int PRE = (MAX_CHUNKS + APPLY_THREADS - 1) / APPLY_THREADS;
    auto dir_row = [&](int task_id) -> const int32_t* {
        int oi = 0;
        while (task_id >= plan.first_task[oi + 1]) ++oi;
        const int lv = plan.order[oi];
        const int sl = (task_id - plan.first_task[oi]) / plan.k_split[lv];
        return ws.dir + ((size_t)lv * MAX_SLICES + sl) * dir_pitch;
    };
    // the first task of every workgroup is its own index (the launch has at most one workgroup per task): no round trip to the
    // queue and no barrier in front of the first directory row; the queue hands out the tasks behind the first round
    const int first_task = task_begin + (int)blockIdx.x;
    if (tid == 0) s_task[0] = first_task;
    if (first_task < task_end) {
        const int32_t* __restrict__ d0 = dir_row(first_task);
        for (int c = tid; c < n_chunks; c += APPLY_THREADS) s_dir2[0][c] = d0[c];
    }
    for (uint32_t k = tid; k < 2 * SLICE2; k += APPLY_THREADS) lds[k] = 0;
    __syncthreads();
    for (int it = 0;; ++it) {
        const int task = s_task[it & 1];
        if (task >= task_end) return;
        int next_task = 0;
        if (tid == 0) next_task = task_begin + (int)gridDim.x + atomicAdd(&ws.queue[group], 1);   // published behind this task's accumulation
#ifdef NGP_BIN_TIMING
        if (tid == 0) ws.timing[4 * task + 0] = (long long)wall_clock64();
#endif
        int oi = 0;
        while (task >= plan.first_task[oi + 1]) ++oi;
        const int level = plan.order[oi];
        const int K = plan.k_split[level];
        const int t = task - plan.first_task[oi];
        const int slice = t / K, part = t - slice * K;
        const uint32_t res = meta.resolution[level];
        const uint32_t size = meta.offset[level + 1] - meta.offset[level];
        const uint32_t lo = (uint32_t)slice * SLICE2;
        const uint32_t len = min(SLICE2, size - lo);
        const int* s_dir = s_dir2[it & 1];
#ifdef NGP_BIN_TIMING
        if (tid == 0) ws.timing[4 * task + 1] = (long long)wall_clock64();
#endif
        const half2_t* __restrict__ g_level = dfeats + (size_t)level * n_samples;
        const int32_t* __restrict__ pool_level = ws.pool + plan.pool_off[level];
        if (level_is_hashed(res, size)) {
            apply_segments_hashed<NGP_APPLY_B>(lds, lo, len, res, size, meta.scale[level], x, box, g_level, active, pool_level, s_dir, n_chunks, part, K);
        }
        else if (NGP_DENSE_RUNS) apply_segments_dense_runs<NGP_DENSE_B>(lds, lo, len, res, size, meta.scale[level], x, box, g_level, active, pool_level, s_dir, n_chunks, part, K);
        else apply_segments_dense<4>(lds, lo, len, res, size, meta.scale[level], x, box, g_level, active, pool_level, s_dir, n_chunks, part, K);
#if NGP_DIR_BY_WAVE0
        // The next task's directory row is fetched by WAVE 0 alone, as soon as its own rows are done and before the barrier: the global
        // round trip hides in the other waves' remaining rows (the waves of a task finish microseconds apart) instead of sitting in the
        // write-out, where all sixteen waited for it (round 6: 3 us of a 20 us task).  The row goes straight into the idle half of s_dir2.
        if (tid < 64) {
            const int nt0 = __shfl(next_task, 0, 64);                  // (thread 0 holds the id the queue returned)
            if (tid == 0) s_task[(it + 1) & 1] = nt0;
            if (nt0 < task_end) {
                const int32_t* __restrict__ dn = dir_row(nt0);
                int* __restrict__ dst = s_dir2[(it + 1) & 1];
                for (int c0 = 0; c0 < n_chunks; c0 += 64 * 8) {         // eight loads in flight per lane
                    int v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int c = c0 + q * 64 + tid; v[q] = c < n_chunks ? dn[c] : 0; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int c = c0 + q * 64 + tid; if (c < n_chunks) dst[c] = v[q]; }
                }
            }
        }
#else
        if (tid == 0) s_task[(it + 1) & 1] = next_task;
#endif
        lds_barrier();                                                 // accumulators complete, next id visible
#ifdef NGP_BIN_TIMING
        if (tid == 0) ws.timing[4 * task + 2] = (long long)wall_clock64();
#endif
#if !NGP_DIR_BY_WAVE0
        const int nt = s_task[(it + 1) & 1];
        int pre[PRE];
        if (nt < task_end) {                                           // (uniform) the next task's directory row, in flight under the write-out
            const int32_t* __restrict__ dn = dir_row(nt);
#pragma unroll
            for (int q = 0; q < PRE; ++q) { const int c = tid + q * APPLY_THREADS; pre[q] = c < n_chunks ? dn[c] : 0; }
        }
#endif
        half2_t* __restrict__ out = grad_table + meta.offset[level] + lo;
        const float inv = 1.0f / FIX_SCALE;
#if NGP_APPLY_WRITEOUT_BATCHED
        {
            // all of a thread's accumulators are read (16-byte LDS reads), then cleared, then converted and stored: the reads of one
            // entry do not wait behind the clear of the previous one (same array: the compiler keeps them in program order)
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            constexpr int WO = (int)((SLICE2 + APPLY_THREADS - 1) / APPLY_THREADS), WB = NGP_APPLY_WRITEOUT_BATCHED;
            float2* __restrict__ pout = ws.partial + plan.part_off[level] + (size_t)part * size + lo;
            const ll2 z2 = {0, 0};
#pragma unroll
            for (int q0 = 0; q0 < WO; q0 += WB) {
                ll2 acc2[WB];
#pragma unroll
                for (int q = 0; q < WB; ++q) {
                    const uint32_t k = tid + (q0 + q) * APPLY_THREADS;
                    acc2[q] = *reinterpret_cast<const ll2*>(lds + 2 * (k < len ? k : 0));
                }
#pragma unroll
                for (int q = 0; q < WB; ++q) {
                    const uint32_t k = tid + (q0 + q) * APPLY_THREADS;
                    if (k < len) *reinterpret_cast<ll2*>(lds + 2 * k) = z2;               // ready for the next task
                }
#pragma unroll
                for (int q = 0; q < WB; ++q) {
                    const uint32_t k = tid + (q0 + q) * APPLY_THREADS;
                    if (k < len) {
                        const float a0 = (float)acc2[q][0] * inv, a1 = (float)acc2[q][1] * inv;
                        if (K == 1) { half2_t v; v[0] = sat_f16(a0); v[1] = sat_f16(a1); out[k] = v; }
                        else pout[k] = make_float2(a0, a1);
                    }
                }
            }
        }
#else
        for (uint32_t k = tid; k < len; k += APPLY_THREADS) {
            const float a0 = (float)lds[2 * k] * inv, a1 = (float)lds[2 * k + 1] * inv;
            lds[2 * k] = 0; lds[2 * k + 1] = 0;                        // ready for the next task
            if (K == 1) { half2_t v; v[0] = sat_f16(a0); v[1] = sat_f16(a1); out[k] = v; }
            else ws.partial[plan.part_off[level] + (size_t)part * size + lo + k] = make_float2(a0, a1);
        }
#endif
#if !NGP_DIR_BY_WAVE0
        if (nt < task_end) {
#pragma unroll
            for (int q = 0; q < PRE; ++q) { const int c = tid + q * APPLY_THREADS; if (c < n_chunks) s_dir2[(it + 1) & 1][c] = pre[q]; }
        }
#endif
        lds_barrier();                                                 // accumulators clear, the next task's directory row in place (the write-out's stores still in flight)
#ifdef NGP_BIN_TIMING
        if (tid == 0) ws.timing[4 * task + 3] = (long long)wall_clock64();
#endif
    }
}

// K-split levels: grad[i] = sum over the K partial tables, in part order (deterministic, no atomics).
__global__ void __launch_bounds__(256)
merge_kernel(GridMeta meta, BinPlan plan, BinWs ws, half2_t* __restrict__ grad_table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;        // entry index inside the span of K-split levels
    uint32_t base = 0;
    for (int l = 0; l < meta.n_levels; ++l) {
        if (plan.k_split[l] <= 1) continue;
        const uint32_t size = meta.offset[l + 1] - meta.offset[l];
        if (i - base < size) {
            const uint32_t e = i - base;
            float a = 0.f, b = 0.f;
            for (int p = 0; p < plan.k_split[l]; ++p) {
                const float2 v = ws.partial[plan.part_off[l] + (size_t)p * size + e];
                a += v.x; b += v.y;
            }
            half2_t o; o[0] = sat_f16(a); o[1] = sat_f16(b);
            grad_table[meta.offset[l] + e] = o;
            return;
        }
        base += size;
    }
}

struct BinLayout { size_t queue, timing, dir, pool, partial, bytes; long long merge_entries; };

// Plan: slices of SLICE2 entries; the levels with few slices (coarse, dense) split their lists over
// K tasks so that every level yields at least ~16 tasks.
bool make_plan(const ngp_grid_meta* meta, int n_samples, BinPlan& P, BinLayout& L) {
    P.n_levels = meta->n_levels;
    P.n_chunks = ngp_div_up(n_samples > 0 ? n_samples : 1, CHUNK);
    double cost[NGP_MAX_LEVELS];
    for (int l = 0; l < NGP_MAX_LEVELS; ++l) { P.n_slices[l] = 0; P.k_split[l] = 1; P.order[l] = l; cost[l] = 0; }
    for (int l = 0; l < meta->n_levels; ++l) {
        const uint32_t size = meta->offset[l + 1] - meta->offset[l], res = meta->resolution[l];
        const bool hashed = (uint64_t)res * res * res > size;
        const int ns = (int)((size + SLICE2 - 1) / SLICE2);
        P.n_slices[l] = ns;
#ifndef NGP_DENSE_K_BIG
#define NGP_DENSE_K_BIG 4
#endif
        P.k_split[l] = hashed ? 1 : (ns == 1 ? 16 : ns == 2 ? 8 : NGP_DENSE_K_BIG);      // dense: a z-slab can hold most of the scene's samples (64/ns tasks measured slower)
        // relative cost of one task: a dense slice (a slab of the grid) gets all 8 corners of its samples, a hashed one ~2
        cost[l] = hashed ? 1.0 : 4.0;
    }
    for (int a = 0; a < meta->n_levels; ++a)      // order levels by decreasing task cost (stable)
        for (int b = a + 1; b < meta->n_levels; ++b)
            if (cost[P.order[b]] > cost[P.order[a]]) { const int t = P.order[a]; P.order[a] = P.order[b]; P.order[b] = t; }
    int nt = 0;
    for (int a = 0; a < meta->n_levels; ++a) { P.first_task[a] = nt; nt += P.n_slices[P.order[a]] * P.k_split[P.order[a]]; }
    for (int a = meta->n_levels; a <= NGP_MAX_LEVELS; ++a) P.first_task[a] = nt;
    P.n_tasks = nt;
    long long part_entries = 0; L.merge_entries = 0;
    for (int l = 0; l < NGP_MAX_LEVELS; ++l) P.part_off[l] = 0;
    for (int l = 0; l < meta->n_levels; ++l) {
        if (P.k_split[l] <= 1) continue;
        const long long size = meta->offset[l + 1] - meta->offset[l];
        P.part_off[l] = part_entries;
        part_entries += size * P.k_split[l];
        L.merge_entries += size;
    }
    long long pool_words = 0;
    for (int l = 0; l < NGP_MAX_LEVELS; ++l) { P.pool_off[l] = 0; P.entry_words[l] = 1; }
    for (int l = 0; l < meta->n_levels; ++l) {
        const uint32_t size = meta->offset[l + 1] - meta->offset[l], res = meta->resolution[l];
        const bool hashed = (uint64_t)res * res * res > size;
        P.entry_words[l] = 1;
        P.pool_off[l] = pool_words;
        pool_words += (long long)P.n_chunks * CHUNK_SLOTS * P.entry_words[l];
    }
    L.queue = 0;
    L.timing = 256;                               // 2048 tasks x 4 x 8 B (NGP_BIN_TIMING builds)
    L.dir = 256 + 65536;
    L.pool = L.dir + ((size_t)meta->n_levels * MAX_SLICES * P.n_chunks * 4 + 255) / 256 * 256;
    L.partial = L.pool + ((size_t)pool_words * 4 + 255) / 256 * 256;
    L.bytes = L.partial + (size_t)part_entries * sizeof(float2);
    return P.n_chunks <= MAX_CHUNKS;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

size_t ngp_hashgrid_bwd_binned_workspace_bytes(const ngp_grid_meta* meta, int n_samples) {
    if (!meta || n_samples < 0 || meta->n_features != 2 || meta->n_levels < 1 || meta->n_levels > NGP_MAX_LEVELS) return 0;
    BinPlan P; BinLayout L;
    if (!make_plan(meta, n_samples, P, L)) return 0;               // more than MAX_CHUNKS * 1024 samples: use ngp_hashgrid_bwd_sliced
    return L.bytes;
}

// Launch groups for a table backward whose result is handed on piecewise (multi-GPU: the gradient exchange of a finished
// piece runs underneath the slice owners of the next): group 0 = the K-split (dense, coarse) levels + the first hashed
// levels, the remaining hashed levels are spread evenly over the other groups.  Levels are contiguous in the table, so
// every group completes one contiguous range of entries.
static void group_bounds(const ngp_grid_meta* meta, const BinPlan& P, int n_groups, int group, int& order_begin, int& order_end) {
    int n_dense = 0;
    while (n_dense < meta->n_levels && P.k_split[P.order[n_dense]] > 1) ++n_dense;
    const int n_hashed = meta->n_levels - n_dense;
    // hashed levels per group: as even as possible, group 0 takes the remainder together with the dense levels' (cheap) tables
    auto first_hashed = [&](int g) { return g <= 0 ? 0 : (int)(((long long)n_hashed * g + n_groups - 1) / n_groups); };
    order_begin = group == 0 ? 0 : n_dense + first_hashed(group);
    order_end = n_dense + (group + 1 >= n_groups ? n_hashed : first_hashed(group + 1));
}

int ngp_hashgrid_bwd_binned_group_entries(const ngp_grid_meta* meta, int n_samples, int n_groups, int group,
                                          int64_t* entry_begin, int64_t* entry_end) {
    if (!meta || n_groups < 1 || n_groups > 16 || group < 0 || group >= n_groups || !entry_begin || !entry_end) return NGP_EINVAL;
    BinPlan P; BinLayout L;
    if (!make_plan(meta, n_samples > 0 ? n_samples : 1, P, L)) return NGP_EUNSUP;
    int a, b;
    group_bounds(meta, P, n_groups, group, a, b);
    for (int i = 0; i + 1 < meta->n_levels; ++i) if (P.order[i] > P.order[i + 1]) return NGP_EUNSUP;     // levels in table order (coarse = dense first)
    *entry_begin = a < meta->n_levels ? meta->offset[P.order[a]] : meta->offset[meta->n_levels];
    *entry_end = b < meta->n_levels ? meta->offset[P.order[b]] : meta->offset[meta->n_levels];
    return 0;
}

// partials_out != NULL (ngp_hashgrid_bwd_binned_deferred): the merge of the K-split levels' partial tables is left to the consumer
// of the gradient (the fused Adam reads the K partials itself) and *partials_out says where they are.
static int binned_group_impl(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* dfeats,
                             const ngp_grid_meta* meta, int n_samples, const int32_t* active_idx,
                             const int32_t* n_active, void* workspace, size_t workspace_bytes,
                             ngp_half* grad_table, int n_groups, int group, ngp_grid_partials* partials_out, ngp_stream_t stream) {
    if (n_samples < 0 || !meta || meta->n_features != 2 || meta->n_levels < 1 || meta->n_levels > NGP_MAX_LEVELS) return NGP_EINVAL;
    if (n_groups < 1 || n_groups > 16 || group < 0 || group >= n_groups) return NGP_EINVAL;
    NGP_CHECK_PTR(workspace);
    NGP_CHECK_PTR(grad_table);
    if (n_samples > 0) { NGP_CHECK_PTR(x); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(dfeats); }
    if (active_idx != nullptr && n_active == nullptr) return NGP_EINVAL;      // n_active alone: x and dfeats both in compact order
    BinPlan P; BinLayout L;
    if (!make_plan(meta, n_samples, P, L)) return NGP_EUNSUP;
    if (workspace_bytes < L.bytes) return NGP_EINVAL;
    for (int l = 0; l < meta->n_levels; ++l) {
        const uint32_t size = meta->offset[l + 1] - meta->offset[l], res = meta->resolution[l];
        if ((uint64_t)res * res * res > size && (size & (size - 1)) != 0) return NGP_EUNSUP;   // corner_indices masks instead of %
        if (P.n_slices[l] > MAX_SLICES) return NGP_EUNSUP;
    }
    hipStream_t st = ngp_stream(stream);
    char* wsb = static_cast<char*>(workspace);
    BinWs ws;
    ws.queue = reinterpret_cast<int32_t*>(wsb + L.queue);
    ws.timing = reinterpret_cast<long long*>(wsb + L.timing);
    ws.dir = reinterpret_cast<int32_t*>(wsb + L.dir);
    ws.pool = reinterpret_cast<int32_t*>(wsb + L.pool);
    ws.partial = reinterpret_cast<float2*>(wsb + L.partial);
    hipError_t e = hipSuccess;
    const GridMeta dm = to_dev_meta(meta);
    if (group == 0)
        bin_kernel<<<dim3(meta->n_levels * P.n_chunks), dim3(BIN_THREADS), 0, st>>>(
            x, xyz_min, xyz_max, (const half2_t*)dfeats, dm, P, ws, n_samples, active_idx, n_active);
    constexpr int smem = (int)(SLICE2 * 2 * sizeof(long long));
    static bool attr_set[64] = {};              // per device: the attribute belongs to the device's code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(apply_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    int a, b;
    group_bounds(meta, P, n_groups, group, a, b);
    const int task_begin = P.first_task[a], task_end = P.first_task[b];
    const int n_tasks = task_end - task_begin;
    if (n_tasks > 0) {
        const int n_wg = n_tasks < NGP_APPLY_WGS ? n_tasks : NGP_APPLY_WGS;
        apply_kernel<false><<<dim3(n_wg), dim3(APPLY_THREADS), smem, st>>>(
                x, xyz_min, xyz_max, (const half2_t*)dfeats, dm, P, ws, n_samples, active_idx, n_active, (half2_t*)grad_table, group, task_begin, task_end);
    }
    if (partials_out != nullptr) {
        // the K-split levels must be a prefix of the table (coarse levels first: true for every grid this package builds)
        ngp_grid_partials& o = *partials_out;
        int nd = 0;
        while (nd < meta->n_levels && P.k_split[nd] > 1) ++nd;
        for (int l = nd; l < meta->n_levels; ++l) if (P.k_split[l] > 1) return NGP_EUNSUP;
        o.n_levels = nd;
        for (int l = 0; l < NGP_MAX_LEVELS; ++l) { o.k_split[l] = 1; o.part_off[l] = 0; }
        for (int l = 0; l <= NGP_MAX_LEVELS; ++l) o.offset[l] = meta->offset[l < nd ? l : nd];
        for (int l = 0; l < nd; ++l) { o.k_split[l] = P.k_split[l]; o.part_off[l] = P.part_off[l]; }
        o.value_end = 2 * (int64_t)meta->offset[nd];
        o.partial = reinterpret_cast<const float*>(ws.partial);
    } else if (group == 0 && L.merge_entries > 0) {     // the K-split levels all sit in group 0
        merge_kernel<<<dim3(ngp_div_up(L.merge_entries, 256)), dim3(256), 0, st>>>(dm, P, ws, (half2_t*)grad_table);
    }
    return NGP_LAUNCH_RESULT();
}

int ngp_hashgrid_bwd_binned_group(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* dfeats,
                                  const ngp_grid_meta* meta, int n_samples, const int32_t* active_idx,
                                  const int32_t* n_active, void* workspace, size_t workspace_bytes,
                                  ngp_half* grad_table, int n_groups, int group, ngp_stream_t stream) {
    return binned_group_impl(x, xyz_min, xyz_max, dfeats, meta, n_samples, active_idx, n_active, workspace, workspace_bytes, grad_table,
                             n_groups, group, nullptr, stream);
}

int ngp_hashgrid_bwd_binned_deferred(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* dfeats,
                                     const ngp_grid_meta* meta, int n_samples, const int32_t* active_idx,
                                     const int32_t* n_active, void* workspace, size_t workspace_bytes,
                                     ngp_half* grad_table, ngp_grid_partials* partials_out, ngp_stream_t stream) {
    NGP_CHECK_PTR(partials_out);
    return binned_group_impl(x, xyz_min, xyz_max, dfeats, meta, n_samples, active_idx, n_active, workspace, workspace_bytes, grad_table,
                             1, 0, partials_out, stream);
}

int ngp_hashgrid_bwd_binned(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* dfeats,
                            const ngp_grid_meta* meta, int n_samples, const int32_t* active_idx,
                            const int32_t* n_active, void* workspace, size_t workspace_bytes,
                            ngp_half* grad_table, ngp_stream_t stream) {
    return ngp_hashgrid_bwd_binned_group(x, xyz_min, xyz_max, dfeats, meta, n_samples, active_idx, n_active, workspace, workspace_bytes,
                                         grad_table, 1, 0, stream);
}

#pragma GCC visibility pop
}  // extern "C"
