// Multiresolution hash-grid encoding, forward and backward, for gfx950.
//
// Algorithm: tiny-cuda-nn's GridEncoding (NVlabs/tiny-cuda-nn, include/tiny-cuda-nn/encodings/
// grid.h: grid_scale / grid_resolution / pos_fract / grid_index / coherent prime hash), as
// configured by the reference at /root/reference/models/networks.py:36-48 (L=16, F=2, T=2^19,
// N_min=16, linear interpolation).  tiny-cuda-nn is NOT vendored by the reference and its
// version is unpinned; the restatement here and in oracle/tcnn_oracle.py is from its published
// source.
//
// MI355X mapping (DESIGN.md "hash grid"):
//   * work item = (sample, level); one thread gathers the 8 corners of one level (8 x 4 B
//     half2 loads in flight per lane), so a workgroup only ever touches ONE level's table;
//   * workgroups are mapped level-major and XCD-aware: workgroup b runs (observed, speed
//     only) on XCD b%8, and every level with a large table is handed whole to one XCD, so the
//     2 MiB hashed table it is gathering from stays resident in that XCD's private 4 MiB L2
//     instead of all 22.8 MB competing for every L2; the small dense levels fill the XCDs up
//     to equal cost (FwdMap; launches sized for a device-side count pair levels {x, 15-x});
//   * the lanes of a wave are consecutive samples of a ray: only the first lane of a run of
//     lanes in one cell gathers, the rest take the corners from it (cell runs, below);
//   * features are stored LEVEL-MAJOR [L][S] half2: every store/load of the stream is a
//     contiguous 256 B per wave (the row-major (S,32) layout would be 4-byte writes at a
//     64-byte stride);
//   * the per-sample position/feature streams use non-temporal accesses so they do not evict
//     the table from L2;
//   * backward: no global atomics (measured 19 G/s on MI355X whatever the pattern): the
//     gradient table is privatised slice by slice in LDS (see hashgrid_bwd_sliced_kernel).
#include "hashgrid_common.h"
#include <hip/hip_fp16.h>
#include <cstdlib>
#include <cstring>

using namespace ngp_grid;

namespace {

// workgroup -> (level, chunk).  n_chunks = ceil(S/256).
__device__ __forceinline__ bool map_block(int n_levels, int n_chunks, int& level, int& chunk) {
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int slot = q / n_chunks;
    chunk = q - slot * n_chunks;
    if (n_levels == 16) level = (slot == 0) ? xcd : 15 - xcd;   // pair a small dense level with a hashed one
    else level = xcd + 8 * slot;
    return level < n_levels;
}

// Cost-balanced map of the forward (round 3).  The pair map above gives XCD 0 the cheapest and the most expensive level: measured
// one level at a time (305 k ray-ordered samples, tools/profile_fwd_levels.py) a level costs 16 us (res 16) .. 38 us (res 1025),
// the launch takes as long as its slowest pair (0 + 15: 54 us) while the sum over the levels is 340 us, 42 per XCD.  Here every
// level's chunks are dealt out in 16 phases (chunk % 16) and an XCD takes a set of phases per level (make_fwd_map): any prefix of
// the chunks (device-sized batches) is balanced the same way.
struct FwdMap {
    // per XCD up to 16 pieces = (level, set of phases), taken one after the other (a large table is in use by one piece at a
    // time): end[k][j] = number of the XCD's workgroups up to and including piece j, piece[k][j] = level << 16 | phase mask,
    // magic[k][j] = ceil(2^32 / phases of the piece) (division by multiplication).  The row arrives with three scalar loads;
    // measured alternatives: a serial walk over per-level masks in kernel-argument memory (16 dependent loads per workgroup:
    // 1.6x slower), period-major order (chunks in lockstep over all pieces: 47 instead of 45 us, two 2 MiB tables alternate
    // in the L2 of the XCDs that own two).  Launches sized for a bound on a device-side sample count keep the pair map: a
    // workgroup that finds its chunk beyond the count costs ~100 scalar instructions here (the scalar unit is shared by a CU's
    // SIMDs), 5x what it costs there (bound = 2x count: 258 us against 234).
    int32_t end[8][16];
    uint32_t piece[8][16];
    uint32_t magic[8][16];
    int32_t blocks_per_xcd;             // 0: pair map
};

// workgroup b -> (level, chunk) under a piece table (host and device: ngp_debug_hashgrid_fwd_map walks it on the CPU)
__host__ __device__ __forceinline__ bool fwd_map_lookup(const FwdMap& m, int n_chunks, int b, int& level, int& chunk) {
    const int xcd = b & 7, q = b >> 3;
    int e[16]; uint32_t pc[16], mg[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { e[j] = m.end[xcd][j]; pc[j] = m.piece[xcd][j]; mg[j] = m.magic[xcd][j]; }
    if (q >= e[15]) return false;
    int begin = 0; uint32_t mine = pc[0], mag = mg[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) if (q >= e[j - 1]) { begin = e[j - 1]; mine = pc[j]; mag = mg[j]; }
    const uint32_t mk = mine & 0xFFFFu;
    const int per = __builtin_popcount(mk), full = n_chunks >> 4, ql = q - begin;
    const int quot = per == 1 ? ql : (int)(uint32_t)(((unsigned long long)(uint32_t)ql * mag) >> 32);     // ql / per, exact for ql < 2^28
    const int period = quot < full ? quot : full, r = ql - period * per;
    uint32_t t = mk;
    for (int k = 0; k < r; ++k) t &= t - 1u;
    level = (int)(mine >> 16); chunk = period * 16 + (int)__builtin_ctz(t);
    return true;
}
__device__ __forceinline__ bool map_block_weighted(const FwdMap& m, int n_chunks, int& level, int& chunk) {
    return fwd_map_lookup(m, n_chunks, (int)blockIdx.x, level, chunk);
}

__device__ __forceinline__ void cell_of(const float* __restrict__ x, const Box& box, size_t i, float scale,
                                        uint32_t (&p)[3], float (&f)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x01 = (__builtin_nontemporal_load(x + 3 * i + k) - box.mn[k]) * box.inv[k];
        const float pos = fmaf(x01, scale, 0.5f);
        const float fl = floorf(pos);
        p[k] = (uint32_t)(int)fl;
        f[k] = pos - fl;
    }
}

// The 8 corner values of a cell: 8 independent 4-byte gathers.  (Fetching the x-neighbour pairs of a hashed level as one aligned 8-byte
// gather -- 6 lane addresses per cell instead of 8 -- was built and measured in round 4: forward 0.060 -> 0.071 ms, not kept;
// profiles/archive_r01_r04/r04_step_ab.txt (c).)
template <bool HASHED>
__device__ __forceinline__ void gather_corners(const half2_t* __restrict__ tab, const uint32_t (&p)[3], uint32_t res, uint32_t size,
                                               half2_t (&v)[8]) {
    uint32_t idx[8];
    corner_indices<HASHED>(p, res, size, idx);
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = tab[idx[c]];
}

template <bool HASHED>
__device__ __forceinline__ void encode_one(const half2_t* __restrict__ tab, uint32_t res, uint32_t size,
                                           const uint32_t (&p)[3], const float (&f)[3], float& o0, float& o1) {
    half2_t v[8];
    gather_corners<HASHED>(tab, p, res, size, v);
    o0 = 0.f; o1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w = corner_weight(c, f);
        o0 = fmaf(w, (float)v[c][0], o0);
        o1 = fmaf(w, (float)v[c][1], o1);
    }
}

// Cell runs.  The lanes of a wave are consecutive samples of a ray, and on the coarse levels dozens of them sit in ONE cell (a
// level-0 cell is 37 march steps across): 64 lanes then ask for the same 8 corners.  The vector memory path charges per lane
// address whatever the addresses are (measured: the dense levels gather only ~2x faster than the hashed ones), so on levels up
// to reuse_max_res only the FIRST lane of every run of equal cells gathers; the other lanes of the run take the 8 values from it
// through the LDS crossbar (ds_bpermute: 8 per wave instead of 8 x 64 lane addresses).  Same values, same arithmetic: bit-identical.
__device__ __forceinline__ half2_t lane_read(half2_t v, int src_lane) {
    return __builtin_bit_cast(half2_t, __shfl(__builtin_bit_cast(int, v), src_lane, 64));
}

// SPT samples per thread: all 8*SPT gathers of a thread are issued before the first blend.
template <int SPT>
__global__ void __launch_bounds__(256)
hashgrid_fwd_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                    const half2_t* __restrict__ table, GridMeta meta, int n_samples, int n_chunks,
                    const int32_t* __restrict__ n_dev, half2_t* __restrict__ feats, FwdMap fmap) {
    int level, chunk;
    if (fmap.blocks_per_xcd > 0) { if (!map_block_weighted(fmap, n_chunks, level, chunk)) return; }
    else if (!map_block(meta.n_levels, n_chunks, level, chunk)) return;
    // device-sized batches (sync-free callers): the launch covers an upper bound, *n_dev is the
    // real sample count and also the level stride of `feats`
    if (n_dev != nullptr) {
        n_samples = min(*n_dev, n_samples);
        if (chunk * SPT * 256 >= n_samples) return;
    }
    const uint32_t res = meta.resolution[level];
    const uint32_t size = meta.offset[level + 1] - meta.offset[level];
    const half2_t* __restrict__ tab = table + meta.offset[level];
    const Box box = load_box(xyz_min, xyz_max);
    const bool hashed = level_is_hashed(res, size);
    const float scale = meta.scale[level];
    static_assert(SPT == 1, "one sample per thread: 2 or 4 measured the same (the kernel is bound by gather transactions)");
    {
        const int i = chunk * 256 + threadIdx.x, lane = threadIdx.x & 63;
        const bool ok = i < n_samples;
        uint32_t p[3]; float f[3];
        cell_of(x, box, (size_t)(ok ? i : n_samples - 1), scale, p, f);       // lanes past the end: the last sample's cell, no run of their own
        const uint32_t q0 = __shfl_up(p[0], 1, 64), q1 = __shfl_up(p[1], 1, 64), q2 = __shfl_up(p[2], 1, 64);
        const bool first = lane == 0 || p[0] != q0 || p[1] != q1 || p[2] != q2;
        const unsigned long long firsts = __ballot(first);
        const int src = 63 - __builtin_clzll(firsts & ((2ull << lane) - 1ull)); // the run's first lane (lane 0 always is one)
        half2_t v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c][0] = (_Float16)0; v[c][1] = (_Float16)0; }
        if (first) {
            if (hashed) gather_corners<true>(tab, p, res, size, v);
            else gather_corners<false>(tab, p, res, size, v);
        }
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half2_t u = lane_read(v[c], src);
            const float w = corner_weight(c, f);
            o0 = fmaf(w, (float)u[0], o0);
            o1 = fmaf(w, (float)u[1], o1);
        }
        half2_t out; out[0] = (_Float16)o0; out[1] = (_Float16)o1;
        if (ok) __builtin_nontemporal_store(out, feats + (size_t)level * n_samples + i);
    }
}


// The same forward over an explicit LIST of sample ids (the two-round forward of the training step, csrc/stepper.hip): work item j
// of n_list (host bound; *n_list_dev on the device) encodes sample list[j] and writes its features at the sample's own place in the
// level-major array (level stride = n_samples).  Ids come in runs of consecutive samples of a ray, so the streams stay mostly
// coalesced; results are per-sample, i.e. independent of the list order.
__global__ void __launch_bounds__(256)
hashgrid_fwd_list_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                         const half2_t* __restrict__ table, GridMeta meta, int n_samples, const int32_t* __restrict__ list,
                         int n_list, int n_chunks, const int32_t* __restrict__ n_list_dev, half2_t* __restrict__ feats) {
    int level, chunk;
    if (!map_block(meta.n_levels, n_chunks, level, chunk)) return;
    if (n_list_dev != nullptr) n_list = min(*n_list_dev, n_list);
    const uint32_t res = meta.resolution[level];
    const uint32_t size = meta.offset[level + 1] - meta.offset[level];
    const half2_t* __restrict__ tab = table + meta.offset[level];
    const Box box = load_box(xyz_min, xyz_max);
    const bool hashed = level_is_hashed(res, size);
    const float scale = meta.scale[level];
    // the launch has a FIXED number of chunks per level (a list whose length only the device knows would otherwise be launched
    // for its bound: 20 000 workgroups that look at the count and leave cost more than the work); chunks stride over the list
    // cell runs as in hashgrid_fwd_kernel: the list holds runs of consecutive samples of a ray; only the first lane of a run of
    // equal cells gathers.  The loop is wave-uniform (whole waves step over the list) so that every lane takes part in the exchange.
    const int lane = threadIdx.x & 63;
    for (int jw = chunk * 256 + (int)(threadIdx.x & ~63u); jw < n_list; jw += n_chunks * 256) {
        const int j = jw + lane;
        const int i = j < n_list ? list[j] : -1;
        const bool valid = (unsigned)i < (unsigned)n_samples;            // padding entries are -1
        uint32_t p[3]; float f[3];
        cell_of(x, box, (size_t)(valid ? i : 0), scale, p, f);
        const uint32_t q0 = __shfl_up(p[0], 1, 64), q1 = __shfl_up(p[1], 1, 64), q2 = __shfl_up(p[2], 1, 64);
        const bool prev_valid = __shfl_up((int)valid, 1, 64) != 0;
        const bool first = valid && (lane == 0 || !prev_valid || p[0] != q0 || p[1] != q1 || p[2] != q2);
        const unsigned long long firsts = __ballot(first);
        const unsigned long long upto = firsts & ((2ull << lane) - 1ull);
        const int src = upto ? 63 - __builtin_clzll(upto) : lane;        // (a valid lane always finds its run's first lane)
        half2_t v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c][0] = (_Float16)0; v[c][1] = (_Float16)0; }
        if (first) {
            if (hashed) gather_corners<true>(tab, p, res, size, v);
            else gather_corners<false>(tab, p, res, size, v);
        }
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half2_t u = lane_read(v[c], src);
            const float w = corner_weight(c, f);
            o0 = fmaf(w, (float)u[0], o0);
            o1 = fmaf(w, (float)u[1], o1);
        }
        half2_t out; out[0] = (_Float16)o0; out[1] = (_Float16)o1;
        if (valid) __builtin_nontemporal_store(out, feats + (size_t)level * n_samples + i);
    }
}


// (Coarse levels served from tables RESIDENT IN LDS -- north_star's "gather staged through LDS" -- were built and measured in round 2:
// 147 KB of half2 for levels 0-2 in one CU's LDS, a persistent 1024-thread workgroup per CU; slower than the L2 path, which already
// hits 93 % on those tables and keeps 8 workgroups per CU in flight.  profiles/archive_r01_r04/r02_hashgrid_fwd_lds_experiment.txt; removed in round 5.)

// ---- backward w.r.t. the input positions (pose optimisation, train.py:86-89,117-122) -------------
// tiny-cuda-nn's grid backward-input for linear interpolation: d feat / d pos_k =
// sum over corners of (+-1 along k) * (the other two weights) * value; chain: pos = x01*scale + 0.5,
// x01 = (x - min) / (max - min).  One thread per sample walks the levels (this path is off unless
// the rays carry gradients, it is not on the throughput-critical path).
__global__ void __launch_bounds__(256)
hashgrid_bwd_input_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                          const half2_t* __restrict__ table, const half2_t* __restrict__ dfeats, GridMeta meta,
                          int n_samples, float out_scale, float* __restrict__ dL_dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const Box box = load_box(xyz_min, xyz_max);
    float g[3] = {0.f, 0.f, 0.f};
    for (int level = 0; level < meta.n_levels; ++level) {
        const uint32_t res = meta.resolution[level];
        const uint32_t size = meta.offset[level + 1] - meta.offset[level];
        const half2_t* __restrict__ tab = table + meta.offset[level];
        const float scale = meta.scale[level];
        uint32_t p[3], idx[8]; float f[3];
        cell_of(x, box, (size_t)i, scale, p, f);
        if (level_is_hashed(res, size)) corner_indices<true>(p, res, size, idx);
        else corner_indices<false>(p, res, size, idx);
        const half2_t d = dfeats[(size_t)level * n_samples + i];
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { const half2_t t = tab[idx[c]]; v[c] = (float)t[0] * (float)d[0] + (float)t[1] * (float)d[1]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float w = ((c >> k) & 1) ? 1.f : -1.f;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j != k) w *= ((c >> j) & 1) ? f[j] : 1.f - f[j];
                acc = fmaf(w, v[c], acc);
            }
            g[k] = fmaf(acc, scale * box.inv[k], g[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dx[3 * (size_t)i + k] = g[k] * out_scale;
}

// ---- backward, global-atomic flavour (kept for A/B and for accumulate semantics) ----------------
template <bool HASHED, bool F32>
__device__ __forceinline__ void scatter_one(void* __restrict__ grad_level, uint32_t res, uint32_t size,
                                            const uint32_t (&p)[3], const float (&f)[3], float g0, float g1) {
    uint32_t idx[8];
    corner_indices<HASHED>(p, res, size, idx);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w = corner_weight(c, f);
        if (F32) {
            float* g = reinterpret_cast<float*>(grad_level) + 2 * (size_t)idx[c];
            unsafeAtomicAdd(g, w * g0);
            unsafeAtomicAdd(g + 1, w * g1);
        } else {
            __half2* g = reinterpret_cast<__half2*>(grad_level) + idx[c];
            unsafeAtomicAdd(g, __floats2half2_rn(w * g0, w * g1));   // global_atomic_pk_add_f16
        }
    }
}

template <bool F32>
__global__ void __launch_bounds__(256)
hashgrid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                    const half2_t* __restrict__ dfeats, GridMeta meta, int n_samples, int n_chunks,
                    void* __restrict__ grad_table) {
    int level, chunk;
    if (!map_block(meta.n_levels, n_chunks, level, chunk)) return;
    const int i = chunk * 256 + threadIdx.x;
    if (i >= n_samples) return;
    const half2_t g = __builtin_nontemporal_load(dfeats + (size_t)level * n_samples + i);
    const float g0 = (float)g[0], g1 = (float)g[1];
    if (g0 == 0.f && g1 == 0.f) return;   // samples past a ray's early stop carry exact zeros
    const uint32_t res = meta.resolution[level];
    const uint32_t size = meta.offset[level + 1] - meta.offset[level];
    void* grad_level = F32 ? (void*)(reinterpret_cast<float*>(grad_table) + 2 * (size_t)meta.offset[level])
                           : (void*)(reinterpret_cast<__half2*>(grad_table) + meta.offset[level]);
    const Box box = load_box(xyz_min, xyz_max);
    uint32_t p[3]; float f[3];
    cell_of(x, box, (size_t)i, meta.scale[level], p, f);
    if (level_is_hashed(res, size)) scatter_one<true, F32>(grad_level, res, size, p, f, g0, g1);
    else scatter_one<false, F32>(grad_level, res, size, p, f, g0, g1);
}

// ------------------------------------------------------------------------------------------
// Backward without global atomics: LDS-privatised scatter.
//
// Measured on MI355X: device-scope f16x2 atomics retire at ~19 G/s whatever the address pattern
// (they execute below the XCD-private L2s), i.e. 1.7 ms for the 33.5 M corner updates of a
// 262 k-sample batch.  Instead the gradient table is cut into slices of SLICE entries (128 KiB
// of packed-f16 accumulators); one workgroup owns one slice in its CU's LDS, scans ALL samples
// of its level, recomputes the 8 corner indices (cheap VALU) and applies only the updates that
// fall into its slice with ds_pk_add_f16.  The whole 22.9 MB gradient table lives in the
// chip's 40 MB of LDS for the duration of the kernel and is written out once with plain
// coalesced stores -- the output is OVERWRITTEN (no zero-fill, no accumulate).
//
// `active` (optional) lists the samples that can carry gradient (those up to a ray's early
// stop): dfeats columns are then compact (column j belongs to sample active[j]).
// ------------------------------------------------------------------------------------------
// Decomposition (measured on MI355X, tools/lds_atomic_bench.hip: an LDS *float* atomic costs ~3
// cycles per active lane -- integer ones run at full rate -- so a workgroup's time is ~3 cycles x
// the corner updates that land in its slice; the split below equalises that over ~253 CUs):
//   hashed levels : 20 slices of SLICE entries each, every slice owner scans all samples;
//   dense levels  : few slices, but all of a level's updates land in them, so their samples are
//                   ALSO split over K workgroups whose private copies are merged with global
//                   packed-f16 atomics (non-zero words only; <1 M atomics per step in total).
constexpr uint32_t SLICE = 27600;               // entries per workgroup: 19 slices per 2^19-entry level, 107.8 KiB of LDS
struct SlicePlan { int32_t n_slices[NGP_MAX_LEVELS]; int32_t k_split[NGP_MAX_LEVELS]; int32_t run_max_res; };

// Every trip of the scan loop needs g, (active,) and x from global memory before it can do
// anything, and a workgroup only has 4 waves per SIMD to hide that (LDS caps it at one workgroup
// per CU): measured, the loop ran at one global-load round trip (~2800 cycles) per trip.  So each
// thread handles BATCH samples per trip and issues all their loads before touching any of them.
constexpr int BATCH = 4;
struct SampleIn { float g0, g1, x[3]; };
__device__ __forceinline__ SampleIn load_sample(const float* __restrict__ x, const half2_t* __restrict__ g_level,
                                                const int32_t* __restrict__ active, int i, int n) {
    SampleIn s;
    const int ic = min(i, n - 1);                               // clamp: the load is unconditional
    const half2_t g = g_level[ic];
    const size_t src = active ? (size_t)active[ic] : (size_t)ic;
    s.g0 = (i < n) ? (float)g[0] : 0.f; s.g1 = (i < n) ? (float)g[1] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) s.x[k] = x[3 * src + k];       // plain loads: x is re-read by every slice owner, keep it in L2
    return s;
}
template <bool HASHED>
__device__ __forceinline__ void sliced_scan(half2_t* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, float scale,
                                            const float* __restrict__ x, const Box& box,
                                            const half2_t* __restrict__ g_level, const int32_t* __restrict__ active,
                                            int n_begin, int n) {
    if (n <= n_begin) return;
    for (int base = n_begin + threadIdx.x; base < n; base += BATCH * blockDim.x) {
        SampleIn in[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) in[b] = load_sample(x, g_level, active, base + b * (int)blockDim.x, n);
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const float g0 = in[b].g0, g1 = in[b].g1;
            if (g0 == 0.f && g1 == 0.f) continue;
            uint32_t p[3]; float f[3];
            cell_of_loaded(in[b].x, box, scale, p, f);
            uint32_t idx[8];
            corner_indices<HASHED>(p, res, size, idx);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(c, f);
                half2_t v; v[0] = (_Float16)(w * g0); v[1] = (_Float16)(w * g1);
                const uint32_t local = idx[c] - lo;
                if (local < len)
                    __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(lds + local), v);
            }
        }
    }
}

// Dense (coarse) levels: consecutive samples of a ray sit in the same cell for 10-40 steps, so a
// wave of 64 consecutive samples hammers the same 8 LDS words (same-address atomics serialise:
// this made the level-0..2 workgroups 5x slower than the hashed ones).  Here a thread walks RUN
// consecutive samples, accumulates the 8 corner sums in registers while the cell stays the same
// and touches LDS only when the cell changes.
constexpr int RUN = 16;
__device__ __forceinline__ void flush_run(half2_t* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, uint32_t key,
                                          const float (&a0)[8], const float (&a1)[8]) {
    const uint32_t r2 = res * res;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint32_t i = key + (c & 1) + ((c >> 1) & 1) * res + (c >> 2) * r2;
        i = (i >= size) ? i - size : i;
        const uint32_t local = i - lo;
        if (local < SLICE && (a0[c] != 0.f || a1[c] != 0.f)) {
            half2_t v; v[0] = (_Float16)a0[c]; v[1] = (_Float16)a1[c];
            __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(lds + local), v);
        }
    }
}
__device__ __forceinline__ void sliced_scan_dense_runs(half2_t* lds, uint32_t lo, uint32_t len, uint32_t res, uint32_t size, float scale,
                                                       const float* __restrict__ x, const Box& box,
                                                       const half2_t* __restrict__ g_level, const int32_t* __restrict__ active,
                                                       int n_begin, int n) {
    const uint32_t r2 = res * res;
    if (n <= n_begin) return;
    for (int base = n_begin + threadIdx.x * RUN; base < n; base += blockDim.x * RUN) {
        float a0[8], a1[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { a0[c] = 0.f; a1[c] = 0.f; }
        uint32_t cur = 0xFFFFFFFFu;
        for (int half = 0; half < RUN; half += 8) {          // loads of 8 samples in flight at a time
            if (base + half >= n) break;
            SampleIn in[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) in[b] = load_sample(x, g_level, active, base + half + b, n);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float g0 = in[b].g0, g1 = in[b].g1;
                if (g0 == 0.f && g1 == 0.f) continue;
                uint32_t p[3]; float f[3];
                cell_of_loaded(in[b].x, box, scale, p, f);
                const uint32_t key = min(p[0], res - 1u) + min(p[1], res - 1u) * res + min(p[2], res - 1u) * r2;   // border clamp: corner_indices<false>
                if (key != cur) {
                    if (cur != 0xFFFFFFFFu) flush_run(lds, lo, len, res, size, cur, a0, a1);
#pragma unroll
                    for (int c = 0; c < 8; ++c) { a0[c] = 0.f; a1[c] = 0.f; }
                    cur = key;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(c, f);
                    a0[c] = fmaf(w, g0, a0[c]); a1[c] = fmaf(w, g1, a1[c]);
                }
            }
        }
        if (cur != 0xFFFFFFFFu) flush_run(lds, lo, len, res, size, cur, a0, a1);
    }
}

__global__ void __launch_bounds__(1024)
hashgrid_bwd_sliced_kernel(const float* __restrict__ x, const float* __restrict__ xyz_min, const float* __restrict__ xyz_max,
                           const half2_t* __restrict__ dfeats, GridMeta meta, SlicePlan plan, int n_samples,
                           const int32_t* __restrict__ active, const int32_t* __restrict__ n_active,
                           half2_t* __restrict__ grad_table) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half2_t* lds = reinterpret_cast<half2_t*>(smem_raw);
    // workgroup -> (level, slice, sample split)
    int level = 0, rem = (int)blockIdx.x;
    for (; level < meta.n_levels; ++level) {
        const int nb = plan.n_slices[level] * plan.k_split[level];
        if (rem < nb) break;
        rem -= nb;
    }
    if (level >= meta.n_levels) return;
    const int K = plan.k_split[level];
    const int slice = rem / K, part = rem - slice * K;
    const uint32_t res = meta.resolution[level];
    const uint32_t size = meta.offset[level + 1] - meta.offset[level];
    const uint32_t lo = (uint32_t)slice * SLICE;
    const uint32_t len = min(SLICE, size - lo);
    const half2_t z = {0, 0};
    for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) lds[k] = z;
    __syncthreads();
    const int n = (active && n_active) ? min(*n_active, n_samples) : n_samples;
    // sample range of this split, aligned to the run length so runs are never cut
    const int per = ((n + K - 1) / K + RUN - 1) / RUN * RUN;
    const int n_begin = min(part * per, n), n_end = min(n_begin + per, n);
    const Box box = load_box(xyz_min, xyz_max);
    const half2_t* __restrict__ g_level = dfeats + (size_t)level * n_samples;
    if (level_is_hashed(res, size)) sliced_scan<true>(lds, lo, len, res, size, meta.scale[level], x, box, g_level, active, n_begin, n_end);
    else if ((int)res > plan.run_max_res) sliced_scan<false>(lds, lo, len, res, size, meta.scale[level], x, box, g_level, active, n_begin, n_end);
    else sliced_scan_dense_runs(lds, lo, len, res, size, meta.scale[level], x, box, g_level, active, n_begin, n_end);
    __syncthreads();
    half2_t* __restrict__ out = grad_table + meta.offset[level] + lo;
    if (K == 1) {
        for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) out[k] = lds[k];
    } else {   // merge the K private copies (the host zero-filled this level's range)
        for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) {
            const half2_t v = lds[k];
            if (v[0] != (_Float16)0 || v[1] != (_Float16)0) {
                __half2 hv; __builtin_memcpy(&hv, &v, 4);
                unsafeAtomicAdd(reinterpret_cast<__half2*>(out) + k, hv);
            }
        }
    }
}

// Samples that can carry gradient: the first min(N, total+1) of every ray (composite stops a
// ray once T <= threshold; later samples have w = 0 and exactly zero gradient).  Three small
// kernels: per-ray count, single-workgroup scan, wave-per-ray listing (ray order, deterministic).
__global__ void __launch_bounds__(256)
active_count_kernel(const int64_t* __restrict__ rays_a, const int64_t* __restrict__ total_samples, int n_rays,
                    int32_t* __restrict__ n_act) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int N = (int)rays_a[3 * (size_t)r + 2];
    const int tot = (int)total_samples[rays_a[3 * (size_t)r]];
    n_act[r] = min(N, tot + 1);
}
// in place: n_act[r] <- exclusive prefix; *n_active <- total.  One workgroup, tiles of 8192 counts (8 consecutive
// counts per thread, loaded before anything else): the usual batch of 8192 rays is one load latency + one scan.
__global__ void __launch_bounds__(1024)
active_scan_kernel(int32_t* __restrict__ n_act, int n_rays, int32_t* __restrict__ n_active) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_rays; base += 8192) {
        const int i0 = base + 8 * tid;
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i0 + k < n_rays) ? n_act[i0 + k] : 0;
        int run = ngp_block_scan_tile<8>(v, s_wave, &s_carry);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < n_rays) n_act[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();                               // the carry of this tile is visible to the next
    }
    if (tid == 0) *n_active = s_carry;
}
__global__ void __launch_bounds__(256)
active_write_kernel(const int64_t* __restrict__ rays_a, const int64_t* __restrict__ total_samples,
                    const int32_t* __restrict__ offs, int n_rays, int32_t* __restrict__ active) {
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const int start = (int)rays_a[3 * (size_t)r + 1];
    const int N = (int)rays_a[3 * (size_t)r + 2];
    const int tot = (int)total_samples[rays_a[3 * (size_t)r]];
    const int na = min(N, tot + 1), off = offs[r];
    for (int k = lane; k < na; k += 64) active[off + k] = start + k;
}

__global__ void __launch_bounds__(256)
feats_to_rowmajor_kernel(const half2_t* __restrict__ feats, int n_levels, int n_samples, half2_t* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (sample, level), level fastest
    if (t >= (long long)n_samples * n_levels) return;
    const int l = (int)(t % n_levels); const long long s = t / n_levels;
    out[t] = feats[(size_t)l * n_samples + s];
}

__global__ void __launch_bounds__(256)
feats_from_rowmajor_kernel(const half2_t* __restrict__ in, int n_levels, int n_samples, half2_t* __restrict__ feats) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (level, sample), sample fastest
    if (t >= (long long)n_samples * n_levels) return;
    const int l = (int)(t / n_samples); const long long s = t - (long long)l * n_samples;
    feats[t] = in[(size_t)s * n_levels + l];
}

// see FwdMap.  Costs as measured one level at a time with cell runs on (us per 305 k samples on one XCD): 15 + 0.0225 x resolution
// for a level with a large table, 16 for a dense one.  A large table (> 1 MiB) must stay in ONE XCD's 4 MiB L2 (cutting such levels
// across XCDs by weight alone was measured: 54 -> 94..335 us, three 2 MiB tables per L2), so those levels go whole to the least
// loaded XCD, most expensive first; the small dense levels, whose tables fit any L2 many times over, are dealt out in sixteenths
// to fill the XCDs up to the same load.
FwdMap make_fwd_map(const ngp_grid_meta* meta, int n_chunks, bool exact) {
    FwdMap m;
    uint16_t mask[8][NGP_MAX_LEVELS];
    for (int k = 0; k < 8; ++k) for (int l = 0; l < NGP_MAX_LEVELS; ++l) mask[k][l] = 0;
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 16; ++j) { m.end[k][j] = 0; m.piece[k][j] = 1u; m.magic[k][j] = 0xFFFFFFFFu; }
    m.blocks_per_xcd = 0;
    const float small_cost = 16.0f;
    if (!exact || meta->n_levels < 8) return m;
    float load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool big[NGP_MAX_LEVELS], done[NGP_MAX_LEVELS];
    float cost[NGP_MAX_LEVELS];
    for (int l = 0; l < meta->n_levels; ++l) {
        const uint32_t size = meta->offset[l + 1] - meta->offset[l];
        big[l] = (size_t)size * 4 > (1u << 20);
        const float res = meta->resolution[l] < 2048u ? (float)meta->resolution[l] : 2048.f;
        cost[l] = big[l] ? 15.0f + 0.0225f * res : small_cost;
        done[l] = false;
    }
    auto least = [&] { int b = 0; for (int k = 1; k < 8; ++k) if (load[k] < load[b]) b = k; return b; };
    for (;;) {                                                      // large tables: whole levels, most expensive first
        int pick = -1;
        for (int l = 0; l < meta->n_levels; ++l) if (big[l] && !done[l] && (pick < 0 || cost[l] > cost[pick])) pick = l;
        if (pick < 0) break;
        const int k = least();
        mask[k][pick] = 0xFFFFu; load[k] += cost[pick]; done[pick] = true;
    }
    for (int l = meta->n_levels - 1; l >= 0; --l) {                 // small tables: by sixteenths
        if (big[l]) continue;
        for (int ph = 0; ph < 16; ++ph) {
            const int k = least();
            mask[k][l] |= (uint16_t)(1u << ph); load[k] += cost[l] / 16.f;
        }
    }
    const int full = n_chunks >> 4;
    const uint32_t low = (1u << (n_chunks & 15)) - 1u;
    for (int k = 0; k < 8; ++k) {
        int cnt = 0, j = 0;
        for (int l = meta->n_levels - 1; l >= 0; --l) {
            if (!mask[k][l]) continue;
            const int per = __builtin_popcount(mask[k][l]);
            cnt += full * per + __builtin_popcount(mask[k][l] & low);
            m.end[k][j] = cnt; m.piece[k][j] = ((uint32_t)l << 16) | mask[k][l];
            m.magic[k][j] = per == 1 ? 0xFFFFFFFFu : (uint32_t)(((1ull << 32) + per - 1) / per);
            ++j;
        }
        for (; j < 16; ++j) m.end[k][j] = cnt;                      // (unused pieces: empty ranges)
        if (cnt > m.blocks_per_xcd) m.blocks_per_xcd = cnt;
    }
    return m;
}

int n_blocks_for(int n_levels, int n_chunks) {
    const int slots = (n_levels == 16) ? 2 : (n_levels + 7) / 8;
    return 8 * slots * n_chunks;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

// tiny-cuda-nn grid.h, GridEncodingTemplated constructor: per level
//   scale = exp2(l * log2(per_level_scale)) * base_resolution - 1
//   res   = ceil(scale) + 1
//   n     = min(next_multiple(res^3, 8), 2^log2_hashmap_size)      (hash grid type)
int ngp_grid_meta_init(ngp_grid_meta* meta, int n_levels, int n_features, int log2_hashmap_size,
                       int base_resolution, float per_level_scale) {
    if (!meta || n_levels < 1 || n_levels > NGP_MAX_LEVELS || n_features != 2 ||
        log2_hashmap_size < 1 || log2_hashmap_size > 28 || base_resolution < 1) return NGP_EINVAL;
    meta->n_levels = n_levels; meta->n_features = n_features;
    const float log2_pls = log2f(per_level_scale);
    uint32_t off = 0;
    for (int l = 0; l < NGP_MAX_LEVELS; ++l) { meta->offset[l] = 0; meta->resolution[l] = 0; meta->scale[l] = 0.f; }
    for (int l = 0; l < n_levels; ++l) {
        const float scale = exp2f(l * log2_pls) * base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1;
        const uint32_t max_params = 0xFFFFFFFFu / 2;
        uint32_t n = (powf((float)res, 3.f) > (float)max_params) ? max_params : res * res * res;
        n = (n + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (n > cap) n = cap;
        meta->offset[l] = off; meta->resolution[l] = res; meta->scale[l] = scale;
        off += n;
    }
    for (int l = n_levels; l <= NGP_MAX_LEVELS; ++l) meta->offset[l] = off;
    return 0;
}

int ngp_hashgrid_fwd(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* table,
                     const ngp_grid_meta* meta, int n_samples, ngp_half* feats, ngp_stream_t stream) {
    return ngp_hashgrid_fwd_n(x, xyz_min, xyz_max, table, meta, n_samples, nullptr, feats, stream);
}

int ngp_hashgrid_fwd_n(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* table,
                       const ngp_grid_meta* meta, int n_samples, const int32_t* n_dev, ngp_half* feats,
                       ngp_stream_t stream) {
    if (n_samples < 0 || !meta || meta->n_features != 2) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(x); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(table); NGP_CHECK_PTR(feats);
    // SPT = 1: measured on MI355X, 2 or 4 samples per thread change nothing (57 / 55 / 57 us at 303 k coherent samples):
    // the kernel is bound by L2 gather transactions, not by loads in flight per lane.
    const int n_chunks = ngp_div_up(n_samples, 256);
    const FwdMap fmap = make_fwd_map(meta, n_chunks, n_dev == nullptr);
    const int n_blocks = fmap.blocks_per_xcd > 0 ? 8 * fmap.blocks_per_xcd : n_blocks_for(meta->n_levels, n_chunks);
    hipLaunchKernelGGL(hashgrid_fwd_kernel<1>, dim3(n_blocks), dim3(256), 0, ngp_stream(stream),
                       x, xyz_min, xyz_max, (const half2_t*)table, to_dev_meta(meta), n_samples, n_chunks, n_dev, (half2_t*)feats, fmap);
    return NGP_LAUNCH_RESULT();
}

int ngp_debug_hashgrid_fwd_map(const ngp_grid_meta* meta, int n_chunks, int32_t* xcd_level_chunk, int max_blocks) {
    if (!meta || n_chunks < 1 || max_blocks < 0 || (max_blocks > 0 && !xcd_level_chunk)) return NGP_EINVAL;
    const FwdMap m = make_fwd_map(meta, n_chunks, true);
    if (m.blocks_per_xcd <= 0) return 0;                            // the pair map is in use for this table
    const int n_blocks = 8 * m.blocks_per_xcd;
    int n = 0;
    for (int b = 0; b < n_blocks; ++b) {
        int level, chunk;
        if (!fwd_map_lookup(m, n_chunks, b, level, chunk)) continue;
        if (n < max_blocks) { xcd_level_chunk[3 * n] = b & 7; xcd_level_chunk[3 * n + 1] = level; xcd_level_chunk[3 * n + 2] = chunk; }
        ++n;
    }
    return n;
}

int ngp_hashgrid_fwd_list(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* table,
                          const ngp_grid_meta* meta, int n_samples, const int32_t* list, int n_list_max,
                          const int32_t* n_list_dev, ngp_half* feats, ngp_stream_t stream) {
    if (n_samples < 0 || n_list_max < 0 || !meta || meta->n_features != 2) return NGP_EINVAL;
    if (n_samples == 0 || n_list_max == 0) return 0;
    NGP_CHECK_PTR(x); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(table); NGP_CHECK_PTR(feats); NGP_CHECK_PTR(list);
    int n_chunks = ngp_div_up(n_list_max, 256);
    if (n_chunks > 128) n_chunks = 128;              // 2048 workgroups at most: 8 per CU, each striding over the list
    hipLaunchKernelGGL(hashgrid_fwd_list_kernel, dim3(n_blocks_for(meta->n_levels, n_chunks)), dim3(256), 0, ngp_stream(stream),
                       x, xyz_min, xyz_max, (const half2_t*)table, to_dev_meta(meta), n_samples, list, n_list_max, n_chunks, n_list_dev,
                       (half2_t*)feats);
    return NGP_LAUNCH_RESULT();
}

int ngp_hashgrid_bwd_input(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* table,
                           const ngp_half* dfeats, const ngp_grid_meta* meta, int n_samples, float out_scale,
                           float* dL_dx, ngp_stream_t stream) {
    if (n_samples < 0 || !meta || meta->n_features != 2) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(x); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(table); NGP_CHECK_PTR(dfeats); NGP_CHECK_PTR(dL_dx);
    hipLaunchKernelGGL(hashgrid_bwd_input_kernel, dim3(ngp_div_up(n_samples, 256)), dim3(256), 0, ngp_stream(stream),
                       x, xyz_min, xyz_max, (const half2_t*)table, (const half2_t*)dfeats, to_dev_meta(meta), n_samples, out_scale, dL_dx);
    return NGP_LAUNCH_RESULT();
}

int ngp_hashgrid_bwd(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* dfeats,
                     const ngp_grid_meta* meta, int n_samples, void* grad_table, int grad_is_f32,
                     ngp_stream_t stream) {
    if (n_samples < 0 || !meta || meta->n_features != 2) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(x); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(dfeats); NGP_CHECK_PTR(grad_table);
    const int n_chunks = ngp_div_up(n_samples, 256);
    const dim3 grid(n_blocks_for(meta->n_levels, n_chunks)), block(256);
    if (grad_is_f32)
        hipLaunchKernelGGL(hashgrid_bwd_kernel<true>, grid, block, 0, ngp_stream(stream),
                           x, xyz_min, xyz_max, (const half2_t*)dfeats, to_dev_meta(meta), n_samples, n_chunks, grad_table);
    else
        hipLaunchKernelGGL(hashgrid_bwd_kernel<false>, grid, block, 0, ngp_stream(stream),
                           x, xyz_min, xyz_max, (const half2_t*)dfeats, to_dev_meta(meta), n_samples, n_chunks, grad_table);
    return NGP_LAUNCH_RESULT();
}

int ngp_hashgrid_bwd_sliced(const float* x, const float* xyz_min, const float* xyz_max, const ngp_half* dfeats,
                            const ngp_grid_meta* meta, int n_samples, const int32_t* active_idx,
                            const int32_t* n_active, ngp_half* grad_table, ngp_stream_t stream) {
    if (n_samples < 0 || !meta || meta->n_features != 2) return NGP_EINVAL;
    NGP_CHECK_PTR(grad_table);
    if (n_samples > 0) { NGP_CHECK_PTR(x); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(dfeats); }
    if ((active_idx == nullptr) != (n_active == nullptr)) return NGP_EINVAL;
    SlicePlan plan;
    plan.run_max_res = 1 << 20;            // run accumulation on every dense level (per-sample float atomics measured 4x slower there)
    int n_blocks = 0;
    uint32_t zero_lo = 0, zero_hi = 0;   // pending [lo, hi) entry range to zero-fill
    hipStream_t st = ngp_stream(stream);
    for (int l = 0; l < NGP_MAX_LEVELS; ++l) { plan.n_slices[l] = 0; plan.k_split[l] = 1; }
    for (int l = 0; l < meta->n_levels; ++l) {
        const uint32_t size = meta->offset[l + 1] - meta->offset[l], res = meta->resolution[l];
        const bool hashed = (uint64_t)res * res * res > size;
        if (hashed && (size & (size - 1)) != 0) return NGP_EUNSUP;   // corner_indices masks instead of %
        const int ns = (int)((size + SLICE - 1) / SLICE);
        // dense levels concentrate every sample's 8 updates in `ns` workgroups: give them about as
        // many workgroups (ns*K) as half a hashed level has slices
        int K = 1;
        if (!hashed) K = (ns == 1) ? 4 : 2;   // measured: a dense workgroup scanning all samples takes ~1.5x a hashed one
        plan.n_slices[l] = ns; plan.k_split[l] = K;
        n_blocks += ns * K;
        if (K > 1) {   // merged by atomics: needs zeros.  Dense levels are contiguous -> one fill for all of them
            if (zero_lo == zero_hi) zero_lo = meta->offset[l];
            if (meta->offset[l] != zero_hi && zero_lo != meta->offset[l]) {   // not adjacent to the pending range: flush it
                hipError_t e = hipMemsetAsync(reinterpret_cast<char*>(grad_table) + (size_t)zero_lo * 4, 0, (size_t)(zero_hi - zero_lo) * 4, st);
                if (e != hipSuccess) return (int)e;
                zero_lo = meta->offset[l];
            }
            zero_hi = meta->offset[l + 1];
        }
    }
    if (zero_hi > zero_lo) {
        hipError_t e = hipMemsetAsync(reinterpret_cast<char*>(grad_table) + (size_t)zero_lo * 4, 0, (size_t)(zero_hi - zero_lo) * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    constexpr int smem = (int)(SLICE * sizeof(half2_t));
    static bool attr_set[64] = {};              // per device: the attribute belongs to the device's code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hashgrid_bwd_sliced_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hashgrid_bwd_sliced_kernel<<<dim3(n_blocks), dim3(1024), smem, st>>>(
        x, xyz_min, xyz_max, (const half2_t*)dfeats, to_dev_meta(meta), plan, n_samples, active_idx, n_active, (half2_t*)grad_table);
    return NGP_LAUNCH_RESULT();
}

int ngp_active_samples(const int64_t* rays_a, const int64_t* total_samples, int n_rays, int32_t* ray_offsets,
                       int32_t* active_idx, int32_t* n_active, ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(n_active);
    if (n_rays > 0) { NGP_CHECK_PTR(rays_a); NGP_CHECK_PTR(total_samples); NGP_CHECK_PTR(active_idx); NGP_CHECK_PTR(ray_offsets); }
    hipStream_t st = ngp_stream(stream);
    if (n_rays > 0)
        hipLaunchKernelGGL(active_count_kernel, dim3(ngp_div_up(n_rays, 256)), dim3(256), 0, st, rays_a, total_samples, n_rays, ray_offsets);
    hipLaunchKernelGGL(active_scan_kernel, dim3(1), dim3(1024), 0, st, ray_offsets, n_rays, n_active);
    if (n_rays > 0)
        hipLaunchKernelGGL(active_write_kernel, dim3(ngp_div_up((long long)n_rays * 64, 256)), dim3(256), 0, st,
                           rays_a, total_samples, ray_offsets, n_rays, active_idx);
    return NGP_LAUNCH_RESULT();
}

int ngp_active_scan(int32_t* n_active_per_ray, int n_rays, int32_t* n_active, ngp_stream_t stream) {
    if (n_rays < 0) return NGP_EINVAL;
    NGP_CHECK_PTR(n_active);
    if (n_rays > 0) NGP_CHECK_PTR(n_active_per_ray);
    hipLaunchKernelGGL(active_scan_kernel, dim3(1), dim3(1024), 0, ngp_stream(stream), n_active_per_ray, n_rays, n_active);
    return NGP_LAUNCH_RESULT();
}

int ngp_feats_to_rowmajor(const ngp_half* feats, int n_levels, int n_samples, ngp_half* out, ngp_stream_t stream) {
    if (n_samples < 0 || n_levels < 1) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(out);
    hipLaunchKernelGGL(feats_to_rowmajor_kernel, dim3(ngp_div_up((long long)n_samples * n_levels, 256)), dim3(256), 0,
                       ngp_stream(stream), (const half2_t*)feats, n_levels, n_samples, (half2_t*)out);
    return NGP_LAUNCH_RESULT();
}

int ngp_feats_from_rowmajor(const ngp_half* in, int n_levels, int n_samples, ngp_half* feats, ngp_stream_t stream) {
    if (n_samples < 0 || n_levels < 1) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(in);
    hipLaunchKernelGGL(feats_from_rowmajor_kernel, dim3(ngp_div_up((long long)n_samples * n_levels, 256)), dim3(256), 0,
                       ngp_stream(stream), (const half2_t*)in, n_levels, n_samples, (half2_t*)feats);
    return NGP_LAUNCH_RESULT();
}

}  // extern "C"
