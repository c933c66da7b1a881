/*
 * ngp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar CPU restatement of the reference's `vren` CUDA kernels, used only as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * ngp_pl_amd/ may import, link or call it.
 *
 * Every function follows one kernel of /root/reference/models/csrc line by line (cited at the
 * function).  "One CUDA thread" becomes one iteration of a sequential loop over rays, so the
 * two atomicAdd reservations of raymarching_train (raymarching.cu:237-238) are taken in ray
 * order: rays_a rows and sample segments come out ray-ordered (the reference's order is
 * whatever the atomics produced; parity is defined after sorting its rows by ray_idx).
 *
 * PINNING: the reference has no tests or golden vectors.  This file is
 * pinned instead against the reference's OWN kernel source compiled for the CPU
 * (oracle/build_ref.sh -> oracle/_ref/libvren_ref.so, see tests/test_oracle_vs_ref.py): with
 * contraction off in both (g_fma = 0, g++ -ffp-contract=off) every output must agree bit for
 * bit, and with contraction on (g_fma = 1, g++ -mfma -ffp-contract=fast) likewise.
 *
 * Floating point: compile with -ffp-contract=off.  g_fma selects how the two expressions that
 * nvcc's default -fmad=true contracts AND whose value can change are evaluated:
 *    x  = o + t*d            (raymarching.cu:205,246,357)
 *    t1 = t1 + dt*noise      (raymarching.cu:198)
 * g_fma = 1 (default) uses fmaf (what nvcc emits), g_fma = 0 rounds the product first.
 * All other mul+add pairs of the marching maths have a power-of-two factor (exact product).
 * __expf -> expf (tolerance-level difference, documented in the tests).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SQRT3 1.73205080757f

static int g_fma = 1;
void oracle_set_fma(int on) { g_fma = on; }
int oracle_get_fma(void) { return g_fma; }

static inline float madd(float a, float b, float c) {   /* a*b + c under the selected contraction */
    if (g_fma) return fmaf(a, b, c);
    volatile float p = a * b;
    return p + c;
}
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* helper_math.h:280-283 */
static inline float signf_(float x) { return copysignf(1.0f, x); }                      /* raymarching.cu:7 */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* raymarching.cu:11-13 */
static inline float calc_dt(float t, float exp_step_factor, int max_samples, int grid_size, float scale) {
    return clampf(t * exp_step_factor, SQRT3 / max_samples, SQRT3 * 2 * scale / grid_size);
}
/* raymarching.cu:19-23 */
static inline int mip_from_pos(float x, float y, float z, int cascades) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent; frexpf(mx, &exponent);
    return imin(cascades - 1, imax(0, exponent + 1));
}
/* raymarching.cu:29-32 */
static inline int mip_from_dt(float dt, int grid_size, int cascades) {
    int exponent; frexpf(dt * grid_size, &exponent);
    return imin(cascades - 1, imax(0, exponent));
}
/* raymarching.cu:35-60 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t morton3D_invert(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

/* raymarching.cu:62-70 */
void oracle_morton3D(const int32_t* coords, int n, int32_t* indices) {
    for (int i = 0; i < n; ++i)
        indices[i] = (int32_t)morton3D((uint32_t)coords[3 * i], (uint32_t)coords[3 * i + 1], (uint32_t)coords[3 * i + 2]);
}
/* raymarching.cu:90-101 */
void oracle_morton3D_invert(const int32_t* indices, int n, int32_t* coords) {
    for (int i = 0; i < n; ++i) {
        const int ind = indices[i];
        coords[3 * i] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
        coords[3 * i + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
        coords[3 * i + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
    }
}
/* raymarching.cu:122-141 */
void oracle_packbits(const float* density_grid, int n_bytes, float density_threshold, uint8_t* density_bitfield) {
    for (int n = 0; n < n_bytes; ++n) {
        uint8_t bits = 0;
        for (uint8_t i = 0; i < 8; i++)
            bits |= (density_grid[8 * (size_t)n + i] > density_threshold) ? ((uint8_t)1 << i) : 0;
        density_bitfield[n] = bits;
    }
}

/* intersection.cu:5-22 */
static inline void ray_aabb(const float* o, const float* inv_d, const float* c, const float* h, float* t1t2) {
    float tmin[3], tmax[3];
    for (int k = 0; k < 3; ++k) { tmin[k] = (c[k] - h[k] - o[k]) * inv_d[k]; tmax[k] = (c[k] + h[k] - o[k]) * inv_d[k]; }
    const float a0 = fminf(tmin[0], tmax[0]), a1 = fminf(tmin[1], tmax[1]), a2 = fminf(tmin[2], tmax[2]);
    const float b0 = fmaxf(tmin[0], tmax[0]), b1 = fmaxf(tmin[1], tmax[1]), b2 = fmaxf(tmin[2], tmax[2]);
    const float t1 = fmaxf(fmaxf(a0, a1), a2);
    const float t2 = fminf(fminf(b0, b1), b2);
    if (t1 > t2) { t1t2[0] = -1.0f; t1t2[1] = -1.0f; return; }
    t1t2[0] = t1; t1t2[1] = t2;
}
/* intersection.cu:103-121; dot() = a.x*b.x + a.y*b.y + a.z*b.z (helper_math.h) */
static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void ray_sphere(const float* o, const float* d, const float* c, float radius, float* t1t2) {
    const float co[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
    const float a = dot3(d, d);
    const float half_b = dot3(d, co);
    const float cc = dot3(co, co) - radius * radius;
    const float discriminant = half_b * half_b - a * cc;
    if (discriminant < 0) { t1t2[0] = -1.0f; t1t2[1] = -1.0f; return; }
    const float disc_sqrt = sqrtf(discriminant);
    t1t2[0] = (-half_b - disc_sqrt) / a; t1t2[1] = (-half_b + disc_sqrt) / a;
}

/* Host-side epilogue intersection.cu:95-97: sort each row ascending by t1 (stable), gather. */
static void sort_hits(int max_hits, float* row_t, int64_t* row_i) {
    for (int i = 1; i < max_hits; ++i) {
        const float k0 = row_t[2 * i], k1 = row_t[2 * i + 1]; const int64_t ki = row_i[i];
        int j = i - 1;
        while (j >= 0 && row_t[2 * j] > k0) {
            row_t[2 * j + 2] = row_t[2 * j]; row_t[2 * j + 3] = row_t[2 * j + 1]; row_i[j + 1] = row_i[j]; --j;
        }
        row_t[2 * j + 2] = k0; row_t[2 * j + 3] = k1; row_i[j + 1] = ki;
    }
}

/* intersection.cu:25-100.  The (ray, voxel) thread grid is walked voxel-fastest per ray, so
 * atomic slots are handed out in voxel order (one of the orders the GPU may produce). */
void oracle_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* half_sizes,
                               int n_rays, int n_voxels, int max_hits, int32_t* hit_cnt, float* hits_t, int64_t* hits_voxel_idx) {
    for (int r = 0; r < n_rays; ++r) {
        float* row_t = hits_t + (size_t)r * max_hits * 2; int64_t* row_i = hits_voxel_idx + (size_t)r * max_hits;
        for (int k = 0; k < max_hits; ++k) { row_t[2 * k] = -1.0f; row_t[2 * k + 1] = -1.0f; row_i[k] = -1; }
        hit_cnt[r] = 0;
        const float inv_d[3] = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
        for (int v = 0; v < n_voxels; ++v) {
            float t1t2[2];
            ray_aabb(rays_o + 3 * r, inv_d, centers + 3 * v, half_sizes + 3 * v, t1t2);
            if (t1t2[1] > 0) {
                const int cnt = hit_cnt[r]++;
                if (cnt < max_hits) { row_t[2 * cnt] = fmaxf(t1t2[0], 0.0f); row_t[2 * cnt + 1] = t1t2[1]; row_i[cnt] = v; }
            }
        }
        sort_hits(max_hits, row_t, row_i);
    }
}
/* intersection.cu:124-197 */
void oracle_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* radii,
                                 int n_rays, int n_spheres, int max_hits, int32_t* hit_cnt, float* hits_t, int64_t* hits_sphere_idx) {
    for (int r = 0; r < n_rays; ++r) {
        float* row_t = hits_t + (size_t)r * max_hits * 2; int64_t* row_i = hits_sphere_idx + (size_t)r * max_hits;
        for (int k = 0; k < max_hits; ++k) { row_t[2 * k] = -1.0f; row_t[2 * k + 1] = -1.0f; row_i[k] = -1; }
        hit_cnt[r] = 0;
        for (int s = 0; s < n_spheres; ++s) {
            float t1t2[2];
            ray_sphere(rays_o + 3 * r, rays_d + 3 * r, centers + 3 * s, radii[s], t1t2);
            if (t1t2[1] > 0) {
                const int cnt = hit_cnt[r]++;
                if (cnt < max_hits) { row_t[2 * cnt] = fmaxf(t1t2[0], 0.0f); row_t[2 * cnt + 1] = t1t2[1]; row_i[cnt] = s; }
            }
        }
        sort_hits(max_hits, row_t, row_i);
    }
}

/* One step of the marching loop body, raymarching.cu:205-233 (identical text at :246-278 and
 * :357-401).  scale_dt is the value handed to calc_dt's `scale` (the test kernel hands it
 * `cascades`, raymarching.cu:370,399).  Returns occupancy; *t is advanced only when empty. */
static inline int march_step(const float* o, const float* d, const float* d_inv, const uint8_t* bitfield,
                             int cascades, int grid_size, float scale, float scale_dt, float esf, int max_samples,
                             float* t, float* xyz, float* dt_out) {
    const uint32_t grid_size3 = (uint32_t)(grid_size * grid_size * grid_size);
    const float grid_size_inv = 1.0f / grid_size;
    const float x = madd(*t, d[0], o[0]), y = madd(*t, d[1], o[1]), z = madd(*t, d[2], o[2]);
    const float dt = calc_dt(*t, esf, max_samples, grid_size, scale_dt);
    const int mip = imax(mip_from_pos(x, y, z, cascades), mip_from_dt(dt, grid_size, cascades));
    const float mip_bound = fminf(scalbnf(1.0f, mip - 1), scale);
    const float mip_bound_inv = 1 / mip_bound;
    const int nx = (int)clampf(0.5f * (x * mip_bound_inv + 1) * grid_size, 0.0f, grid_size - 1.0f);
    const int ny = (int)clampf(0.5f * (y * mip_bound_inv + 1) * grid_size, 0.0f, grid_size - 1.0f);
    const int nz = (int)clampf(0.5f * (z * mip_bound_inv + 1) * grid_size, 0.0f, grid_size - 1.0f);
    const uint32_t idx = mip * grid_size3 + morton3D(nx, ny, nz);
    const int occ = (bitfield[idx / 8] & (1 << (idx % 8))) != 0;
    xyz[0] = x; xyz[1] = y; xyz[2] = z; *dt_out = dt;
    if (!occ) {
        const float tx = (((nx + 0.5f + 0.5f * signf_(d[0])) * grid_size_inv * 2 - 1) * mip_bound - x) * d_inv[0];
        const float ty = (((ny + 0.5f + 0.5f * signf_(d[1])) * grid_size_inv * 2 - 1) * mip_bound - y) * d_inv[1];
        const float tz = (((nz + 0.5f + 0.5f * signf_(d[2])) * grid_size_inv * 2 - 1) * mip_bound - z) * d_inv[2];
        const float t_target = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do { *t += calc_dt(*t, esf, max_samples, grid_size, scale_dt); } while (*t < t_target);
    }
    return occ;
}

/* raymarching.cu:166-332.  Buffers xyzs/dirs (cap,3), deltas/ts (cap) are caller-provided with
 * cap >= total samples (the reference allocates N_rays*max_samples).  Returns total samples, or
 * -1 if cap is too small.  counter[0] = S, counter[1] = R. */
long long oracle_raymarching_train(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                                   int cascades, float scale, float exp_step_factor, const float* noise, int grid_size,
                                   int max_samples, int n_rays, long long cap,
                                   int64_t* rays_a, float* xyzs, float* dirs, float* deltas, float* ts, int32_t* counter) {
    counter[0] = 0; counter[1] = 0;
    for (int r = 0; r < n_rays; ++r) {
        const float* o = rays_o + 3 * r; const float* d = rays_d + 3 * r;
        const float d_inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float t1 = hits_t[2 * r], t2 = hits_t[2 * r + 1];
        if (t1 >= 0) {   /* only perturb the starting t */
            const float dt = calc_dt(t1, exp_step_factor, max_samples, grid_size, scale);
            t1 = madd(dt, noise[r], t1);
        }
        /* first pass: count */
        float t = t1; int N_samples = 0;
        while (0 <= t && t < t2 && N_samples < max_samples) {
            float xyz[3], dt;
            if (march_step(o, d, d_inv, density_bitfield, cascades, grid_size, scale, scale, exp_step_factor, max_samples, &t, xyz, &dt)) {
                t += dt; N_samples++;
            }
        }
        /* second pass: write */
        const int start_idx = counter[0]; counter[0] += N_samples;
        const int ray_count = counter[1]; counter[1] += 1;
        if ((long long)counter[0] > cap) return -1;
        rays_a[3 * (size_t)ray_count] = r; rays_a[3 * (size_t)ray_count + 1] = start_idx; rays_a[3 * (size_t)ray_count + 2] = N_samples;
        t = t1; int samples = 0;
        while (t < t2 && samples < N_samples) {
            float xyz[3], dt;
            const float t_here = t;
            if (march_step(o, d, d_inv, density_bitfield, cascades, grid_size, scale, scale, exp_step_factor, max_samples, &t, xyz, &dt)) {
                const size_t s = (size_t)start_idx + samples;
                xyzs[3 * s] = xyz[0]; xyzs[3 * s + 1] = xyz[1]; xyzs[3 * s + 2] = xyz[2];
                dirs[3 * s] = d[0]; dirs[3 * s + 1] = d[1]; dirs[3 * s + 2] = d[2];
                ts[s] = t_here; deltas[s] = dt;
                t += dt; samples++;
            }
        }
    }
    return counter[0];
}

/* raymarching.cu:335-454.  Outputs (n_alive, n_samples, .) are zero-filled first (:420-425). */
void oracle_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive_indices,
                             const uint8_t* density_bitfield, int cascades, float scale, float exp_step_factor,
                             int grid_size, int max_samples, int N_samples, int n_alive,
                             float* xyzs, float* dirs, float* deltas, float* ts, int32_t* N_eff_samples) {
    memset(xyzs, 0, sizeof(float) * 3 * (size_t)n_alive * N_samples);
    memset(dirs, 0, sizeof(float) * 3 * (size_t)n_alive * N_samples);
    memset(deltas, 0, sizeof(float) * (size_t)n_alive * N_samples);
    memset(ts, 0, sizeof(float) * (size_t)n_alive * N_samples);
    for (int n = 0; n < n_alive; ++n) {
        const size_t r = (size_t)alive_indices[n];
        const float* o = rays_o + 3 * r; const float* d = rays_d + 3 * r;
        const float d_inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float t = hits_t[2 * r], t2 = hits_t[2 * r + 1];
        int s = 0;
        while (t < t2 && s < N_samples) {
            float xyz[3], dt;
            const float t_here = t;
            /* calc_dt is called with `cascades` in the scale slot (raymarching.cu:370,399) */
            if (march_step(o, d, d_inv, density_bitfield, cascades, grid_size, scale, (float)cascades, exp_step_factor, max_samples, &t, xyz, &dt)) {
                const size_t q = (size_t)n * N_samples + s;
                xyzs[3 * q] = xyz[0]; xyzs[3 * q + 1] = xyz[1]; xyzs[3 * q + 2] = xyz[2];
                dirs[3 * q] = d[0]; dirs[3 * q + 1] = d[1]; dirs[3 * q + 2] = d[2];
                ts[q] = t_here; deltas[q] = dt;
                t += dt;
                hits_t[2 * r] = t;
                s++;
            }
        }
        N_eff_samples[n] = s;
    }
}

/* volumerendering.cu:6-84 (outputs zero-initialised as the host wrapper does, :58-62) */
void oracle_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                               const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                               int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws) {
    memset(opacity, 0, sizeof(float) * n_rays); memset(depth, 0, sizeof(float) * n_rays);
    memset(rgb, 0, sizeof(float) * 3 * n_rays); memset(ws, 0, sizeof(float) * (size_t)n_samples);
    memset(total_samples, 0, sizeof(int64_t) * n_rays);
    for (int n = 0; n < n_rays; ++n) {
        const int ray_idx = (int)rays_a[3 * n], start_idx = (int)rays_a[3 * n + 1], N_samples = (int)rays_a[3 * n + 2];
        int samples = 0; float T = 1.0f;
        while (samples < N_samples) {
            const int s = start_idx + samples;
            const float a = 1.0f - expf(-sigmas[s] * deltas[s]);
            const float w = a * T;
            rgb[3 * ray_idx] += w * rgbs[3 * s]; rgb[3 * ray_idx + 1] += w * rgbs[3 * s + 1]; rgb[3 * ray_idx + 2] += w * rgbs[3 * s + 2];
            depth[ray_idx] += w * ts[s];
            opacity[ray_idx] += w;
            ws[s] = w;
            T *= 1.0f - a;
            if (T <= T_threshold) break;
            samples++;
        }
        total_samples[ray_idx] = samples;
    }
}

/* volumerendering.cu:87-202.  dL_dws may be NULL (zeros). */
void oracle_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                               const float* sigmas, const float* rgbs, const float* ws, const float* deltas, const float* ts,
                               const int64_t* rays_a, const float* opacity, const float* depth, const float* rgb,
                               float T_threshold, int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs) {
    memset(dL_dsigmas, 0, sizeof(float) * (size_t)n_samples); memset(dL_drgbs, 0, sizeof(float) * 3 * (size_t)n_samples);
    float* dL_dws_times_ws = (float*)malloc(sizeof(float) * (size_t)(n_samples > 0 ? n_samples : 1));
    for (int s = 0; s < n_samples; ++s) dL_dws_times_ws[s] = (dL_dws ? dL_dws[s] : 0.0f) * ws[s];   /* :175 */
    for (int n = 0; n < n_rays; ++n) {
        const int ray_idx = (int)rays_a[3 * n], start_idx = (int)rays_a[3 * n + 1], N_samples = (int)rays_a[3 * n + 2];
        int samples = 0;
        const float R = rgb[3 * ray_idx], G = rgb[3 * ray_idx + 1], B = rgb[3 * ray_idx + 2];
        const float O = opacity[ray_idx], D = depth[ray_idx];
        float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, d = 0.0f;
        if (N_samples <= 0) continue;   /* the reference would read index start-1 here (:123); nothing is written */
        for (int k = 1; k < N_samples; ++k) dL_dws_times_ws[start_idx + k] += dL_dws_times_ws[start_idx + k - 1];   /* inclusive_scan */
        const float dL_dws_times_ws_sum = dL_dws_times_ws[start_idx + N_samples - 1];
        while (samples < N_samples) {
            const int s = start_idx + samples;
            const float a = 1.0f - expf(-sigmas[s] * deltas[s]);
            const float w = a * T;
            r += w * rgbs[3 * s]; g += w * rgbs[3 * s + 1]; b += w * rgbs[3 * s + 2];
            d += w * ts[s];
            T *= 1.0f - a;
            dL_drgbs[3 * s] = dL_drgb[3 * ray_idx] * w;
            dL_drgbs[3 * s + 1] = dL_drgb[3 * ray_idx + 1] * w;
            dL_drgbs[3 * s + 2] = dL_drgb[3 * ray_idx + 2] * w;
            dL_dsigmas[s] = deltas[s] * (
                dL_drgb[3 * ray_idx] * (rgbs[3 * s] * T - (R - r)) +
                dL_drgb[3 * ray_idx + 1] * (rgbs[3 * s + 1] * T - (G - g)) +
                dL_drgb[3 * ray_idx + 2] * (rgbs[3 * s + 2] * T - (B - b)) +
                dL_dopacity[ray_idx] * (1 - O) +
                dL_ddepth[ray_idx] * (ts[s] * T - (D - d)) +
                T * (dL_dws ? dL_dws[s] : 0.0f) - (dL_dws_times_ws_sum - dL_dws_times_ws[s]));
            if (T <= T_threshold) break;
            samples++;
        }
    }
    free(dL_dws_times_ws);
}

/* volumerendering.cu:205-285 */
void oracle_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                              int64_t* alive_indices, float T_threshold, const int32_t* N_eff_samples,
                              int n_alive, int n_samples, float* opacity, float* depth, float* rgb) {
    for (int n = 0; n < n_alive; ++n) {
        if (N_eff_samples[n] == 0) { alive_indices[n] = -1; continue; }
        const size_t r = (size_t)alive_indices[n];
        int s = 0; float T = 1 - opacity[r];
        while (s < N_eff_samples[n]) {
            const size_t q = (size_t)n * n_samples + s;
            const float a = 1.0f - expf(-sigmas[q] * deltas[q]);
            const float w = a * T;
            rgb[3 * r] += w * rgbs[3 * q]; rgb[3 * r + 1] += w * rgbs[3 * q + 1]; rgb[3 * r + 2] += w * rgbs[3 * q + 2];
            depth[r] += w * ts[q];
            opacity[r] += w;
            T *= 1.0f - a;
            if (T <= T_threshold) { alive_indices[n] = -1; break; }
            s++;
        }
    }
}

/* losses.cu:9-109 */
void oracle_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a,
                               int n_rays, int n_samples, float* loss, float* ws_inclusive_scan, float* wts_inclusive_scan) {
    float* wts = (float*)malloc(sizeof(float) * (size_t)(n_samples + 1));
    float* ws_ex = (float*)calloc((size_t)(n_samples + 1), sizeof(float));
    float* wts_ex = (float*)calloc((size_t)(n_samples + 1), sizeof(float));
    float* _loss = (float*)malloc(sizeof(float) * (size_t)(n_samples + 1));
    for (int s = 0; s < n_samples; ++s) { wts[s] = ws[s] * ts[s]; ws_inclusive_scan[s] = 0; wts_inclusive_scan[s] = 0; }
    for (int n = 0; n < n_rays; ++n) {
        const int start_idx = (int)rays_a[3 * n + 1], N_samples = (int)rays_a[3 * n + 2];
        float a = 0, b = 0;
        for (int k = 0; k < N_samples; ++k) {
            const int s = start_idx + k;
            ws_ex[s] = a; wts_ex[s] = b;
            a += ws[s]; b += wts[s];
            ws_inclusive_scan[s] = a; wts_inclusive_scan[s] = b;
        }
    }
    for (int s = 0; s < n_samples; ++s)   /* :94-95, ATen elementwise */
        _loss[s] = 2 * (wts_inclusive_scan[s] * ws_ex[s] - ws_inclusive_scan[s] * wts_ex[s]) + 1.0f / 3 * ws[s] * ws[s] * deltas[s];
    for (int n = 0; n < n_rays; ++n) loss[n] = 0;
    for (int n = 0; n < n_rays; ++n) {
        const int ray_idx = (int)rays_a[3 * n], start_idx = (int)rays_a[3 * n + 1], N_samples = (int)rays_a[3 * n + 2];
        float acc = 0;
        for (int k = 0; k < N_samples; ++k) acc += _loss[start_idx + k];
        loss[ray_idx] = acc;
    }
    free(wts); free(ws_ex); free(wts_ex); free(_loss);
}

/* losses.cu:112-175 */
void oracle_distortion_loss_bw(const float* dL_dloss, const float* ws_inclusive_scan, const float* wts_inclusive_scan,
                               const float* ws, const float* deltas, const float* ts, const int64_t* rays_a,
                               int n_rays, int n_samples, float* dL_dws) {
    memset(dL_dws, 0, sizeof(float) * (size_t)n_samples);
    for (int n = 0; n < n_rays; ++n) {
        const int ray_idx = (int)rays_a[3 * n], start_idx = (int)rays_a[3 * n + 1], N_samples = (int)rays_a[3 * n + 2];
        const int end_idx = start_idx + N_samples - 1;
        if (N_samples <= 0) continue;
        const float ws_sum = ws_inclusive_scan[end_idx];
        const float wts_sum = wts_inclusive_scan[end_idx];
        for (int s = start_idx; s <= end_idx; s++) {
            dL_dws[s] = dL_dloss[ray_idx] * 2 * (
                (s == start_idx ? (float)0 : (ts[s] * ws_inclusive_scan[s - 1] - wts_inclusive_scan[s - 1])) +
                (wts_sum - wts_inclusive_scan[s] - ts[s] * (ws_sum - ws_inclusive_scan[s])));
            dL_dws[s] += dL_dloss[ray_idx] * (float)2 / 3 * ws[s] * deltas[s];
        }
    }
}
