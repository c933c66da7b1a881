// Shared device helpers for the gfx950 Instant-NGP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ngp_hip.h"
#include "ngp_internal.h"

#define NGP_WAVE 64

#define NGP_CHECK_PTR(p) do { if ((p) == nullptr) return NGP_EINVAL; } while (0)
#define NGP_LAUNCH_RESULT() ((int)hipGetLastError())

static inline hipStream_t ngp_stream(ngp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ngp_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- Morton code (3 x 10 bit), semantics of raymarching.cu:35-60 ----
__device__ __forceinline__ uint32_t ngp_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t ngp_morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return ngp_expand_bits(x) | (ngp_expand_bits(y) << 1) | (ngp_expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t ngp_compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xC30C30C3u;
    x = (x | (x >> 4)) & 0x0F00F00Fu;
    x = (x | (x >> 8)) & 0xFF0000FFu;
    x = (x | (x >> 16)) & 0x0000FFFFu;
    return x;
}

// ---- counter-based RNG (PCG output function on a Weyl-style counter) ----
__device__ __forceinline__ uint32_t ngp_pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// Key of item i under a 64-bit seed: the three words are hashed in turn, NOT folded into one 32-bit base that the item index is
// added to -- with `hash(seed) + i` two launches whose bases land within n of each other draw shifted copies of one sequence
// (about 10^3 such pairs over a 30 000-step run of 8192-ray marches).  Draw j of item i = ngp_pcg_hash(key + j).
__device__ __forceinline__ uint32_t ngp_rng_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t i) {
    return ngp_pcg_hash(ngp_pcg_hash(seed_lo ^ ngp_pcg_hash(i)) + seed_hi);
}

// ---- wave64 helpers ----
__device__ __forceinline__ float ngp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int ngp_wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exclusive scan across a workgroup (up to 1024 threads) where every thread owns ITEMS consecutive counts (v, already
// loaded: all of a tile's loads are in flight together, one memory latency per tile of blockDim.x*ITEMS counts).
// Returns the exclusive prefix of the thread's first item, carry-in included; *s_carry is advanced by the tile
// total.  s_wave: 16 ints, s_carry: 1 int of LDS (zeroed by the caller before the first tile).
template <int ITEMS>
__device__ __forceinline__ int ngp_block_scan_tile(const int (&v)[ITEMS], int* s_wave, int* s_carry) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) mine += v[k];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = *s_carry;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    __syncthreads();                                   // everyone has read the carry and the wave totals
    if (tid == (int)blockDim.x - 1) *s_carry = off + incl;
    return off + incl - mine;
}
