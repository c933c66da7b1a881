// Pieces of the multiresolution hash grid shared by the forward/backward translation units
// (tiny-cuda-nn grid.h semantics as configured at /root/reference/models/networks.py:36-48).
#pragma once
#include "ngp_common.h"

namespace ngp_grid {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr uint32_t PRIME_Y = 2654435761u, PRIME_Z = 805459861u;

struct GridMeta {
    int32_t n_levels;
    uint32_t offset[NGP_MAX_LEVELS + 1];
    uint32_t resolution[NGP_MAX_LEVELS];
    float scale[NGP_MAX_LEVELS];
};

// x01 = (x - min) * (1/(max - min))  [networks.py:103 divides: identical bits for the power-of-two extents 2 x scale of every recipe
// of the reference (scale 0.5 ... 16); for any other extent v_rcp_f32 is 1 ulp off and the product within 2 ulp of the quotient --
// INTEGRATION.md "Box extents"], pos = x01*scale + 0.5, cell = floor(pos).
struct Box { float mn[3], inv[3]; };
#ifndef NGP_BOX_EXACT_RECIPROCAL
#define NGP_BOX_EXACT_RECIPROCAL 0
#endif
__device__ __forceinline__ Box load_box(const float* __restrict__ xyz_min, const float* __restrict__ xyz_max) {
    Box b;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = xyz_min[k];
#if NGP_BOX_EXACT_RECIPROCAL
        // correctly rounded reciprocal for extents that are not powers of two, behind a scalar branch (A/B builds)
        const uint32_t eb = __builtin_amdgcn_readfirstlane(__float_as_uint(xyz_max[k] - xyz_min[k]));
        const float e = __uint_as_float(eb);
        if ((eb & 0x007fffffu) == 0u) b.inv[k] = __builtin_amdgcn_rcpf(e);
        else b.inv[k] = 1.0f / e;
#else
        // v_rcp_f32: exact for a power of two, 1 ulp otherwise (INTEGRATION.md "Box extents")
        b.inv[k] = __builtin_amdgcn_rcpf(xyz_max[k] - xyz_min[k]);
#endif
    }
    return b;
}
// tiny-cuda-nn grid_index(): the dense stride walk uses the hash iff res^3 overflows the level
__device__ __forceinline__ bool level_is_hashed(uint32_t res, uint32_t size) {
    uint32_t stride = 1;
    for (int d = 0; d < 3 && stride <= size; ++d) stride *= res;
    return size < stride;
}

// The 8 corner indices of a cell.  Hashed levels always have a power-of-two size (the cap
// 2^log2_hashmap_size), so `% size` is a mask; dense indices of an in-box cell stay below 2*size,
// so `% size` is one conditional subtract.  Positions OUTSIDE the box (public NGP.density()/forward()
// callers: mesh extraction, user grids; the marcher never emits them) have cell coordinates below 0
// (wrapped to huge unsigned values) or above res-1: tiny-cuda-nn's full `%` returns an arbitrary
// but in-bounds entry for them, here the cell is clamped to the border cell (in bounds as well;
// the weights are left as computed, like tiny-cuda-nn's).
template <bool HASHED>
__device__ __forceinline__ void corner_indices(const uint32_t (&p)[3], uint32_t res, uint32_t size, uint32_t (&idx)[8]) {
    if (HASHED) {
        const uint32_t hx[2] = {p[0], p[0] + 1u};
        const uint32_t hy0 = p[1] * PRIME_Y, hz0 = p[2] * PRIME_Z;
        const uint32_t hy[2] = {hy0, hy0 + PRIME_Y}, hz[2] = {hz0, hz0 + PRIME_Z};
        const uint32_t mask = size - 1u;
#pragma unroll
        for (int c = 0; c < 8; ++c) idx[c] = (hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[c >> 2]) & mask;
    } else {
        const uint32_t r2 = res * res, top = res - 1u;
        const uint32_t base = min(p[0], top) + min(p[1], top) * res + min(p[2], top) * r2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t i = base + (c & 1) + ((c >> 1) & 1) * res + (c >> 2) * r2;
            idx[c] = (i >= size) ? i - size : i;
        }
    }
}
__device__ __forceinline__ float corner_weight(int c, const float (&f)[3]) {
    return ((c & 1) ? f[0] : 1.f - f[0]) * (((c >> 1) & 1) ? f[1] : 1.f - f[1]) * ((c >> 2) ? f[2] : 1.f - f[2]);
}

__device__ __forceinline__ void cell_of_loaded(const float (&xin)[3], const Box& box, float scale, uint32_t (&p)[3], float (&f)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pos = fmaf((xin[k] - box.mn[k]) * box.inv[k], scale, 0.5f);
        const float fl = floorf(pos);
        p[k] = (uint32_t)(int)fl;
        f[k] = pos - fl;
    }
}

inline GridMeta to_dev_meta(const ngp_grid_meta* m) {
    GridMeta d;
    d.n_levels = m->n_levels;
    for (int l = 0; l < NGP_MAX_LEVELS; ++l) {
        d.offset[l] = m->offset[l]; d.resolution[l] = m->resolution[l]; d.scale[l] = m->scale[l];
    }
    d.offset[NGP_MAX_LEVELS] = m->offset[NGP_MAX_LEVELS];
    return d;
}

}  // namespace ngp_grid
