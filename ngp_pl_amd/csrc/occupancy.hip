// Occupancy-grid maintenance on the device (reference: NGP.update_density_grid and its helpers,
// /root/reference/models/networks.py:155-195, 240-269; train.py:160-163 calls it every 16 steps).
//
// The reference does this with ~100 small torch launches and two host syncs (nonzero(), .item()).
// Here one call enqueues, per cascade: occupancy words + prefix (2 launches), the cell draws (1), their regrouping
// by Morton block with the jittered positions (3; warm-up needs none: every cell once, in order), the density-only
// field forward (hash grid + density MLP whose epilogue scatters sigma straight into the scratch grid, 2), then the
// decay/max merge with the masked mean (1) and the bit packing with the device-side threshold (1).
//
// Sampling semantics (networks.py:169-195): per cascade M = G^3/4 cells uniform over the grid plus
// M cells uniform over {cell : density_grid > density_threshold} (with replacement); warm-up: every
// cell once (networks.py:155-167).  The occupied draw is an inverse-CDF lookup over the 64-cell
// occupancy words (prefix of popcounts + select-in-word) instead of nonzero()+randint().  An empty
// occupied set maps every draw to the last cell (the reference adds no samples then; the extra
// evaluations only refresh that cell).  Random numbers: counter-based hash keyed by (seed, i).
#include "ngp_common.h"

namespace {

__device__ __forceinline__ uint32_t occ_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// one wave per 64-cell word: occupancy bits + popcount
__global__ void __launch_bounds__(256)
occ_words_kernel(const float* __restrict__ grid, float threshold, int n_words,
                 unsigned long long* __restrict__ words, int32_t* __restrict__ counts) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_words) return;
    const unsigned long long m = __ballot(grid[(size_t)w * 64 + lane] > threshold);
    if (lane == 0) { words[w] = m; counts[w] = __popcll(m); }
}

// exclusive scan of counts in one workgroup; prefix[n] = total.  Each of the 16 waves owns one contiguous run of the counts and
// walks it 256 counts (64 lanes x 16 bytes, coalesced) at a time: pass 1 sums the run, the 16 run totals are exchanged through
// LDS, pass 2 re-reads (L2) and writes the prefixes -- two barriers in all.  (Tile by tile with four barriers per 1024 counts it
// took 36 us for the 32 768 words of a 128^3 grid.)
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__global__ void __launch_bounds__(1024)
occ_scan_kernel(const int32_t* __restrict__ counts, int n, int32_t* __restrict__ prefix) {
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n4 = n >> 2;                                     // whole int4 groups; a tail of n & 3 counts goes to the last thread
    const int run = ((n4 + 15) / 16 + 63) / 64 * 64;           // int4 groups per wave, a multiple of the 64 lanes
    const int begin = min(wave * run, n4), end = min(begin + run, n4);
    const int4* c4 = reinterpret_cast<const int4*>(counts);
    int sum = 0;
    for (int g = begin + lane; g < end; g += 64) { const int4 v = c4[g]; sum += (v.x + v.y) + (v.z + v.w); }
    sum = wave_incl_scan(sum, lane);
    if (lane == 63) s_wave[wave] = sum;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < wave; ++w) carry += s_wave[w];
    for (int g0 = begin; g0 < end; g0 += 64) {
        const int g = g0 + lane;
        int4 v = {0, 0, 0, 0};
        if (g < end) v = c4[g];
        const int local = (v.x + v.y) + (v.z + v.w);
        const int incl = wave_incl_scan(local, lane);
        int4 o; o.x = carry + incl - local; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
        if (g < end) reinterpret_cast<int4*>(prefix)[g] = o;
        carry += __shfl(incl, 63, 64);
    }
    if (tid == 1023) {                                         // the last wave's carry is the total of the int4 part
        int total = carry;
        for (int i = 4 * n4; i < n; ++i) { prefix[i] = total; total += counts[i]; }
        prefix[n] = total;
    }
}

// position of the r-th (0-based) set bit of m
__device__ __forceinline__ int select_bit(unsigned long long m, int r) {
    int pos = 0;
#pragma unroll
    for (int width = 32; width > 0; width >>= 1) {
        const unsigned long long low = (m >> pos) & ((1ull << width) - 1ull);
        const int c = __popcll(low);
        if (r >= c) { r -= c; pos += width; }
    }
    return pos;
}

// cell i of the update -> (Morton index, jittered world position)
//   mode 0 (warm-up): cell i itself, n = G^3
//   mode 1: i < M uniform coordinates (networks.py:181-184), i >= M the occupied draw (:186-192)
__global__ void __launch_bounds__(256)
occ_sample_kernel(int mode, int n, int M, int grid_size, int n_words, const unsigned long long* __restrict__ words,
                  const int32_t* __restrict__ prefix, float s_minus_hgs, float hgs, uint32_t seed_lo, uint32_t seed_hi,
                  int32_t* __restrict__ cell_idx, float* __restrict__ xyzs, float* __restrict__ zero_stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0 && zero_stats != nullptr) { zero_stats[0] = 0.f; zero_stats[1] = 0.f; }   // (an 8-byte hipMemsetAsync is a 7 us launch)
    const uint32_t base = occ_hash(seed_lo ^ occ_hash(seed_hi + 0x9E3779B9u)) + 7u * (uint32_t)i;
    uint32_t idx, cx, cy, cz;
    if (mode == 0) {
        idx = (uint32_t)i;
    } else if (i < M) {
        cx = (uint32_t)(((uint64_t)occ_hash(base) * (uint64_t)grid_size) >> 32);
        cy = (uint32_t)(((uint64_t)occ_hash(base + 1u) * (uint64_t)grid_size) >> 32);
        cz = (uint32_t)(((uint64_t)occ_hash(base + 2u) * (uint64_t)grid_size) >> 32);
        idx = ngp_morton3D(cx, cy, cz);
    } else {
        const int total = prefix[n_words];
        if (total == 0) {
            idx = (uint32_t)n_words * 64u - 1u;
        } else {
            int rank = (int)(u01(occ_hash(base + 3u)) * (float)total);        // (rand * count).int()
            rank = min(rank, total - 1);
            int lo = 0, hi = n_words;                                          // last word with prefix <= rank
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (prefix[mid] <= rank) lo = mid; else hi = mid;
            }
            idx = (uint32_t)lo * 64u + (uint32_t)select_bit(words[lo], rank - prefix[lo]);
        }
    }
    if (mode == 0 || i >= M) {
        cx = ngp_compact_bits(idx); cy = ngp_compact_bits(idx >> 1); cz = ngp_compact_bits(idx >> 2);
    }
    cell_idx[i] = (int32_t)idx;
    if (xyzs == nullptr) return;                 // mode 1: positions are generated where the draws are put in evaluation order
    // (coords/(G-1)*2-1)*(s-hgs) + (rand*2-1)*hgs   (networks.py:253-255), torch op order
    const float gm1 = (float)(grid_size - 1);
    const uint32_t c[3] = {cx, cy, cz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float centre = ((float)c[k] / gm1 * 2.0f - 1.0f) * s_minus_hgs;
        xyzs[3 * (size_t)i + k] = centre + (u01(occ_hash(base + 4u + k)) * 2.0f - 1.0f) * hgs;
    }
}

// ---- evaluation order of the 2M draws of an update ----
// The order in which the update evaluates its draws is free (sigma is scattered by cell index), but the hash forward of a million
// i.i.d. positions gathers incoherently (329 us; ~210 us for the same positions grouped by cell block).  The draws are therefore
// regrouped by the top bits of their Morton index, each half of the update (uniform / occupied, networks.py:181-192) on its own:
// a counting sort with workgroup-private LDS histograms -- no global atomics (a million returning global atomics cost 70 us on
// this part) -- in three small launches: histogram per (workgroup, block), offsets, placement.  Jitter is keyed by the DRAW,
// so the set of evaluated positions does not depend on the order.
constexpr int OCC_BUCKETS = 1024;      // cell blocks per half
constexpr int OCC_SORT_WGS = 64;       // workgroups per half (a multiple of 16).  32 / 64 / 128: histogram + offsets + placement 40 / 38 / 42 us

__global__ void __launch_bounds__(1024)
occ_hist_kernel(const int32_t* __restrict__ drawn, int M, int shift, int32_t* __restrict__ hist) {
    __shared__ int s_hist[OCC_BUCKETS];
    const int half = blockIdx.x / OCC_SORT_WGS, w = blockIdx.x % OCC_SORT_WGS;
    const int chunk = (M + OCC_SORT_WGS - 1) / OCC_SORT_WGS;
    const int begin = min(w * chunk, M), end = min(begin + chunk, M);
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = begin + threadIdx.x; i < end; i += 1024) atomicAdd(&s_hist[((uint32_t)drawn[(size_t)half * M + i]) >> shift], 1);
    __syncthreads();
    hist[(size_t)blockIdx.x * OCC_BUCKETS + threadIdx.x] = s_hist[threadIdx.x];          // [half][workgroup][block]
}

// hist[half][w][b] -> first evaluation slot of workgroup w's draws of block b: half * M + (draws of blocks < b) + (block b draws of
// workgroups < w).  One workgroup per half, thread b walks its column (coalesced across the wave; a row per thread is not: 53 us).
__global__ void __launch_bounds__(1024)
occ_hist_scan_kernel(int32_t* __restrict__ hist, int M) {
    __shared__ int s_wave[16];
    const int b = threadIdx.x, lane = b & 63, wave = b >> 6;
    int32_t* h = hist + (size_t)blockIdx.x * OCC_SORT_WGS * OCC_BUCKETS;
    int total = 0;
#pragma unroll 16
    for (int w = 0; w < OCC_SORT_WGS; ++w) total += h[w * OCC_BUCKETS + b];
    const int incl = wave_incl_scan(total, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int run = (int)blockIdx.x * M + incl - total;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    for (int w0 = 0; w0 < OCC_SORT_WGS; w0 += 16) {          // second walk of the column (L2), 16 loads in flight: counts -> first slots
        int c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = h[(w0 + k) * OCC_BUCKETS + b];
#pragma unroll
        for (int k = 0; k < 16; ++k) { h[(w0 + k) * OCC_BUCKETS + b] = run; run += c[k]; }
    }
}

__global__ void __launch_bounds__(1024)
occ_place_kernel(const int32_t* __restrict__ drawn, int M, int shift, const int32_t* __restrict__ offsets, int grid_size,
                 float s_minus_hgs, float hgs, uint32_t seed_lo, uint32_t seed_hi, int32_t* __restrict__ cell_idx, float* __restrict__ xyzs) {
    __shared__ int s_cursor[OCC_BUCKETS];
    const int half = blockIdx.x / OCC_SORT_WGS, w = blockIdx.x % OCC_SORT_WGS;
    const int chunk = (M + OCC_SORT_WGS - 1) / OCC_SORT_WGS;
    const int begin = min(w * chunk, M), end = min(begin + chunk, M);
    s_cursor[threadIdx.x] = offsets[(size_t)blockIdx.x * OCC_BUCKETS + threadIdx.x];
    __syncthreads();
    const uint32_t seed = occ_hash(seed_lo ^ occ_hash(seed_hi + 0x9E3779B9u));
    const float gm1 = (float)(grid_size - 1);
    for (int k = begin + threadIdx.x; k < end; k += 1024) {
        const int i = half * M + k;                                    // the draw
        const uint32_t idx = (uint32_t)drawn[i];
        const int slot = atomicAdd(&s_cursor[idx >> shift], 1);
        cell_idx[slot] = (int32_t)idx;
        // (coords/(G-1)*2-1)*(s-hgs) + (rand*2-1)*hgs   (networks.py:253-255), torch op order; same jitter stream as occ_sample_kernel
        const uint32_t base = seed + 7u * (uint32_t)i;
        const uint32_t c[3] = {ngp_compact_bits(idx), ngp_compact_bits(idx >> 1), ngp_compact_bits(idx >> 2)};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float centre = ((float)c[a] / gm1 * 2.0f - 1.0f) * s_minus_hgs;
            xyzs[3 * (size_t)slot + a] = centre + (u01(occ_hash(base + 4u + a)) * 2.0f - 1.0f) * hgs;
        }
    }
}

struct OccLayout { size_t tmp, words, counts, prefix, idx, xyzs, feats, stats, drawn, hist, bytes; };
OccLayout occ_layout(int cascades, int grid_size) {
    OccLayout L;
    const size_t cells = (size_t)grid_size * grid_size * grid_size, n_words = cells / 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.tmp = take((size_t)cascades * cells * 4);
    L.words = take(n_words * 8); L.counts = take(n_words * 4); L.prefix = take((n_words + 1) * 4);
    L.idx = take(cells * 4); L.xyzs = take(cells * 12); L.feats = take(cells * 64); L.stats = take(8);
    L.drawn = take(cells * 2);                                 // the 2M = cells / 2 draws in draw order (`idx`: in evaluation order)
    L.hist = take((size_t)2 * OCC_SORT_WGS * OCC_BUCKETS * 4);
    L.bytes = off;
    return L;
}


// ---- NGP.mark_invisible_cells (networks.py:197-238), once before training (train.py:155-158) ---------------------------------
// One thread per cell of every cascade (Morton index = position in the grid): the cell centre is taken to every training camera
// -- p_cam = R^T (x - t) from the camera-to-world pose [R | t], (u, v, d) = K p_cam -- and the cell is VALID when at least one camera
// has it inside its image at depth >= near and NO camera has it inside its image closer than near.  count_grid = fraction of the
// cameras that cover the cell (the erode decay's exponent, networks.py:262-264), density_grid = 0 / -1.  The poses pass through LDS
// MARK_CAMS at a time (48 B per camera), every thread walks them in the same order: no divergence but the two flags, and no
// bound on the number of training cameras (the reference's loop has none either).
constexpr int MARK_CAMS = 1024;
__global__ void __launch_bounds__(256)
mark_invisible_kernel(const float* __restrict__ K, const float* __restrict__ poses, int n_cams, float img_w, float img_h,
                      float near_distance, int grid_size, float scale, int cells, int n_total,
                      float* __restrict__ count_grid, float* __restrict__ density_grid) {
    __shared__ float s_cam[12 * MARK_CAMS];                // per camera: the 9 entries of R^T, then -R^T t
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < n_total;
    const int gc = live ? g : n_total - 1;
    const int c = gc / cells, m = gc - c * cells;
    // s = min(2^(c-1), scale); cell centres at (coord / (G - 1) * 2 - 1) * (s - s / G)   (networks.py:216-221)
    float s = c == 0 ? 0.5f : (float)(1 << (c - 1));
    s = fminf(s, scale);
    const float span = s - s / (float)grid_size, gm1 = (float)(grid_size - 1);
    const float x = ((float)ngp_compact_bits((uint32_t)m) / gm1 * 2.0f - 1.0f) * span;
    const float y = ((float)ngp_compact_bits((uint32_t)m >> 1) / gm1 * 2.0f - 1.0f) * span;
    const float z = ((float)ngp_compact_bits((uint32_t)m >> 2) / gm1 * 2.0f - 1.0f) * span;
    const float k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[3], k11 = K[4], k12 = K[5], k20 = K[6], k21 = K[7], k22 = K[8];
    int covered = 0; bool too_near = false;
    for (int c0 = 0; c0 < n_cams; c0 += MARK_CAMS) {
        const int nc = min(MARK_CAMS, n_cams - c0);
        if (c0 > 0) __syncthreads();                       // the previous tile has been read by every thread
        for (int i = threadIdx.x; i < nc; i += blockDim.x) {
            const float* P = poses + 12 * (size_t)(c0 + i);    // row-major 3 x 4: [R | t]
            float* o = s_cam + 12 * i;
            const float t0 = P[3], t1 = P[7], t2 = P[11];
#pragma unroll
            for (int r = 0; r < 3; ++r) {                  // row r of R^T = column r of R
                const float a = P[r], b = P[4 + r], cc = P[8 + r];
                o[3 * r] = a; o[3 * r + 1] = b; o[3 * r + 2] = cc;
                o[9 + r] = -(a * t0 + b * t1 + cc * t2);
            }
        }
        __syncthreads();
        for (int i = 0; i < nc; ++i) {
            const float* o = s_cam + 12 * i;
            const float px = o[0] * x + o[1] * y + o[2] * z + o[9];
            const float py = o[3] * x + o[4] * y + o[5] * z + o[10];
            const float pz = o[6] * x + o[7] * y + o[8] * z + o[11];
            const float ud = k00 * px + k01 * py + k02 * pz, vd = k10 * px + k11 * py + k12 * pz, d = k20 * px + k21 * py + k22 * pz;
            const float u = ud / d, v = vd / d;            // (d == 0: inf / nan, every comparison below is false like torch's)
            const bool in_image = d >= 0.f && u >= 0.f && u < img_w && v >= 0.f && v < img_h;
            covered += (in_image && d >= near_distance) ? 1 : 0;
            too_near = too_near || (in_image && d < near_distance);
        }
    }
    if (!live) return;
    const float count = (float)covered / (float)n_cams;
    count_grid[g] = count;
    density_grid[g] = (count > 0.f && !too_near) ? 0.f : -1.f;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

size_t ngp_occupancy_update_workspace_bytes(int cascades, int grid_size) {
    if (cascades < 1 || grid_size < 4 || grid_size > 256 || (grid_size & (grid_size - 1))) return 0;
    return occ_layout(cascades, grid_size).bytes;
}

int ngp_occupancy_update_workspace_layout(int cascades, int grid_size, size_t* tmp_off, size_t* cell_idx_off, size_t* xyzs_off) {
    if (cascades < 1 || grid_size < 4 || grid_size > 256 || (grid_size & (grid_size - 1))) return NGP_EINVAL;
    NGP_CHECK_PTR(tmp_off); NGP_CHECK_PTR(cell_idx_off); NGP_CHECK_PTR(xyzs_off);
    const OccLayout L = occ_layout(cascades, grid_size);
    *tmp_off = L.tmp; *cell_idx_off = L.idx; *xyzs_off = L.xyzs;
    return 0;
}

// (Split into draw / finish with the NEXT update's draws made ahead on a third stream: built and measured in round 4 -- timed windows
// +0.6 %, the 30 000-step run -8 %: the two cross-stream hand-overs per update cost more than the 0.09 ms they hide; removed in round 5.
// profiles/archive_r01_r04/r04_step_ab.txt (e).)
static int occupancy_impl(float* density_grid, uint8_t* density_bitfield, int cascades, int grid_size, float scale,
                          float density_threshold, float decay, const float* decay_grid, int warmup, uint64_t seed,
                          const float* xyz_min, const float* xyz_max, const ngp_half* table, const ngp_grid_meta* meta,
                          const ngp_half* density_w, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    if (cascades < 1 || grid_size < 4 || grid_size > 256 || (grid_size & (grid_size - 1))) return NGP_EINVAL;
    NGP_CHECK_PTR(density_grid); NGP_CHECK_PTR(workspace);
    if (!meta) return NGP_EINVAL;
    NGP_CHECK_PTR(density_bitfield); NGP_CHECK_PTR(xyz_min); NGP_CHECK_PTR(xyz_max); NGP_CHECK_PTR(table); NGP_CHECK_PTR(density_w);
    const OccLayout L = occ_layout(cascades, grid_size);
    if (workspace_bytes < L.bytes) return NGP_EINVAL;
    char* ws = static_cast<char*>(workspace);
    float* tmp = reinterpret_cast<float*>(ws + L.tmp);
    unsigned long long* words = reinterpret_cast<unsigned long long*>(ws + L.words);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + L.counts);
    int32_t* prefix = reinterpret_cast<int32_t*>(ws + L.prefix);
    int32_t* idx = reinterpret_cast<int32_t*>(ws + L.idx);
    float* xyzs = reinterpret_cast<float*>(ws + L.xyzs);
    ngp_half* feats = reinterpret_cast<ngp_half*>(ws + L.feats);
    float* stats = reinterpret_cast<float*>(ws + L.stats);
    hipStream_t st = ngp_stream(stream);
    const int cells = grid_size * grid_size * grid_size, n_words = cells / 64;
    const int M = cells / 4;                                   // networks.py:254
    const int n = warmup ? cells : 2 * M;
    {
        const hipError_t e = hipMemsetAsync(tmp, 0, (size_t)cascades * cells * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    for (int c = 0; c < cascades; ++c) {
        float* grid_c = density_grid + (size_t)c * cells;
        if (!warmup) {
            hipLaunchKernelGGL(occ_words_kernel, dim3(ngp_div_up((long long)n_words * 64, 256)), dim3(256), 0, st,
                               grid_c, density_threshold, n_words, words, counts);
            hipLaunchKernelGGL(occ_scan_kernel, dim3(1), dim3(1024), 0, st, counts, n_words, prefix);
        }
        // s = min(2^(c-1), scale), half_grid_size = s / G in Python doubles (networks.py:251-253)
        double s = 1.0; for (int k = 0; k < c - 1; ++k) s *= 2.0; if (c == 0) s = 0.5;
        if ((double)scale < s) s = (double)scale;
        const double hgs = s / grid_size;
        const uint64_t sd = seed * 0x9E3779B97F4A7C15ull + (uint64_t)c;
        const uint32_t sd_lo = (uint32_t)sd, sd_hi = (uint32_t)(sd >> 32);
        if (warmup) {
            hipLaunchKernelGGL(occ_sample_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, st, 0, n, M, grid_size, n_words,
                               words, prefix, (float)(s - hgs), (float)hgs, sd_lo, sd_hi, idx, xyzs, c == 0 ? stats : (float*)nullptr);
        } else {
            int32_t* drawn = reinterpret_cast<int32_t*>(ws + L.drawn);
            int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
            int bits = 0; while ((1 << bits) < cells) ++bits;
            const int shift = bits > 10 ? bits - 10 : 0;           // OCC_BUCKETS = 2^10 blocks of 2^shift cells
            hipLaunchKernelGGL(occ_sample_kernel, dim3(ngp_div_up(n, 256)), dim3(256), 0, st, 1, n, M, grid_size, n_words,
                               words, prefix, (float)(s - hgs), (float)hgs, sd_lo, sd_hi, drawn, (float*)nullptr, c == 0 ? stats : (float*)nullptr);
            hipLaunchKernelGGL(occ_hist_kernel, dim3(2 * OCC_SORT_WGS), dim3(1024), 0, st, drawn, M, shift, hist);
            hipLaunchKernelGGL(occ_hist_scan_kernel, dim3(2), dim3(1024), 0, st, hist, M);
            hipLaunchKernelGGL(occ_place_kernel, dim3(2 * OCC_SORT_WGS), dim3(1024), 0, st, drawn, M, shift, hist, grid_size,
                               (float)(s - hgs), (float)hgs, sd_lo, sd_hi, idx, xyzs);
        }
        int rc = ngp_hashgrid_fwd(xyzs, xyz_min, xyz_max, table, meta, n, feats, stream);
        if (rc) return rc;
        rc = ngp_density_fwd_scatter(feats, density_w, n, idx, tmp + (size_t)c * cells, stream);
        if (rc) return rc;
    }
    int rc = ngp_density_grid_update(density_grid, tmp, decay_grid, decay, cascades * cells, stats, stream);
    if (rc) return rc;
    return ngp_packbits_auto(density_grid, cascades * cells / 8, stats, density_threshold, density_bitfield, stream);
}

int ngp_occupancy_update(float* density_grid, uint8_t* density_bitfield, int cascades, int grid_size, float scale,
                         float density_threshold, float decay, const float* decay_grid, int warmup, uint64_t seed,
                         const float* xyz_min, const float* xyz_max, const ngp_half* table, const ngp_grid_meta* meta,
                         const ngp_half* density_w, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    return occupancy_impl(density_grid, density_bitfield, cascades, grid_size, scale, density_threshold, decay, decay_grid, warmup, seed,
                          xyz_min, xyz_max, table, meta, density_w, workspace, workspace_bytes, stream);
}


int ngp_mark_invisible_cells(const float* K, const float* poses, int n_cams, int img_w, int img_h, float near_distance,
                             int cascades, int grid_size, float scale, float* count_grid, float* density_grid, ngp_stream_t stream) {
    if (n_cams < 1 || cascades < 1 || grid_size < 2 || grid_size > 1024 || (grid_size & (grid_size - 1)) || img_w < 1 || img_h < 1)
        return NGP_EINVAL;
    NGP_CHECK_PTR(K); NGP_CHECK_PTR(poses); NGP_CHECK_PTR(count_grid); NGP_CHECK_PTR(density_grid);
    const long long cells = (long long)grid_size * grid_size * grid_size, total = cells * cascades;
    if (total > 0x7fffffffLL) return NGP_EINVAL;
    const size_t smem = 0;
    hipLaunchKernelGGL(mark_invisible_kernel, dim3(ngp_div_up(total, 256)), dim3(256), smem, ngp_stream(stream),
                       K, poses, n_cams, (float)img_w, (float)img_h, near_distance, grid_size, scale, (int)cells, (int)total, count_grid, density_grid);
    return NGP_LAUNCH_RESULT();
}

#pragma GCC visibility pop
}  // extern "C"
