/*
 * ngp_hip.h -- C ABI of libngp_hip.so, the MI355X (gfx950) native Instant-NGP hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one
 * function of the reference's two native dependencies for this path:
 *   - the `vren` pybind module        (/root/reference/models/csrc/binding.cpp:234-250), and
 *   - the `tinycudann` torch modules  (call sites /root/reference/models/networks.py:36-92),
 * plus apex FusedAdam (train.py:131).  The reference ABI is torch::Tensor based; this one is
 * plain C: raw DEVICE pointers, explicit sizes, caller-allocated outputs, a hipStream_t passed
 * as void*.  No torch types.  Return value: 0 on success, a positive hipError_t if a launch
 * failed, a negative NGP_E* code for bad arguments.  Nothing here allocates or synchronises
 * unless the comment says so; all work is enqueued on `stream`.
 *
 * Tensors are contiguous, row-major.  f32 = float, f16 = IEEE half (uint16_t in this header),
 * i64 = int64_t.  "R" = rays, "S" = packed samples.
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ngp_stream_t;   /* hipStream_t */
typedef uint16_t ngp_half;    /* IEEE binary16 bits */

#define NGP_EINVAL   (-1)  /* bad argument (null pointer, size out of range) */
#define NGP_EUNSUP   (-2)  /* configuration not supported by the native kernels */
#define NGP_ETIMEOUT (-3)  /* a bounded host wait on a device result ran out (NGP_SPIN_TIMEOUT_S, default 30 s) */
#define NGP_ECOMM    (-4)  /* an RCCL call failed or RCCL could not be loaded: ngp_comm_last_error() has the text */

#define NGP_MAX_LEVELS 16

/* ABI version; bumped when a signature, a record layout or the set of entry points changes (5: round 5's removals + `stage` in
 * ngp_exchange_config; 6: round 6, ngp_stepper_backward_update's step_state, the loss scaler's two entry points).  A binding asserts the number it was written against (ngp_pl_amd/_lib.py: ABI_VERSION). */
int ngp_abi_version(void);
/* Name of the GPU arch the library was built for ("gfx950"). */
const char* ngp_build_arch(void);

/* Termination guards of the marching kernels (no reference counterpart: raymarching.cu:225-232 loops for ever on a ray
 * whose far hit is infinite or whose step is absorbed by rounding).  A tripped guard ends that ray only and is counted on
 * the device: counts4[0] = skips whose smallest step would not move t, [1] = wave-per-ray tile cap, [2] = serial loop
 * iteration cap, [3] = 1 + index of the last ray that ran into [1].  Synchronous (hipMemcpyFromSymbol); reset != 0 clears the counts. */
int ngp_march_guard_read(uint32_t* counts4, int reset);
/* Diagnostics: the first probe that ran into guard [0] since the library was loaded: t, t_target, the three face distances,
 * ray origin (3), ray direction (3), the smallest step.  Synchronous. */
int ngp_march_guard_first(float* probe12);

/* ------------------------------------------------------------------------------------------
 * vren: intersection            (reference: models/csrc/intersection.cu)
 * ------------------------------------------------------------------------------------------ */

/* vren.ray_aabb_intersect (binding.cpp:4-16, intersection.cu:5-100).
 * rays_o,rays_d (R,3) f32; centers,half_sizes (V,3) f32.
 * out: hit_cnt (R) i32, hits_t (R,max_hits,2) f32, hits_voxel_idx (R,max_hits) i64.
 * Rows are sorted ascending by t1 with unfilled (-1) slots first, as torch::sort leaves them
 * (intersection.cu:95-97). */
int ngp_ray_aabb_intersect(const float* rays_o, const float* rays_d,
                           const float* centers, const float* half_sizes,
                           int n_rays, int n_voxels, int max_hits,
                           int32_t* hit_cnt, float* hits_t, int64_t* hits_voxel_idx,
                           ngp_stream_t stream);

/* vren.ray_sphere_intersect (binding.cpp:19-31, intersection.cu:103-197). radii (V) f32. */
int ngp_ray_sphere_intersect(const float* rays_o, const float* rays_d,
                             const float* centers, const float* radii,
                             int n_rays, int n_spheres, int max_hits,
                             int32_t* hit_cnt, float* hits_t, int64_t* hits_sphere_idx,
                             ngp_stream_t stream);

/* Hot-path specialisation of render()'s prologue (rendering.py:27-29): one box, max_hits=1,
 * and the near-plane clamp hits_t[0] in [0,near) -> near fused in.  hits_t (R,2). */
int ngp_ray_aabb_near(const float* rays_o, const float* rays_d,
                      const float* center, const float* half_size, float near_distance,
                      int n_rays, float* hits_t, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * vren: occupancy grid helpers  (reference: models/csrc/raymarching.cu:35-161)
 * ------------------------------------------------------------------------------------------ */

/* vren.morton3D (binding.cpp:46-50): coords (N,3) i32 -> indices (N) i32. */
int ngp_morton3D(const int32_t* coords, int n, int32_t* indices, ngp_stream_t stream);
/* vren.morton3D_invert (binding.cpp:53-57): indices (N) i32 -> coords (N,3) i32. */
int ngp_morton3D_invert(const int32_t* indices, int n, int32_t* coords, ngp_stream_t stream);
/* vren.packbits (binding.cpp:34-43): bit i of byte n = grid[8n+i] > threshold.
 * grid_is_half: 0 -> f32 grid, 1 -> f16 grid.  n_bytes = C*G^3/8. */
int ngp_packbits(const void* density_grid, int grid_is_half, int n_bytes,
                 float density_threshold, uint8_t* density_bitfield, ngp_stream_t stream);

/* Fused occupancy-grid maintenance (networks.py:248-268), device side only:
 *   grid = (grid<0) ? grid : max(grid*decay, tmp)       [decay scalar, or per cell if decay_grid]
 * and the masked sum/count of grid>0 into stats[0] (f32 sum) / stats[1] (f32 count).
 * stats must be zeroed by the caller. */
int ngp_density_grid_update(float* density_grid, const float* density_grid_tmp,
                            const float* decay_grid /* may be NULL */, float decay,
                            int n_cells, float* stats, ngp_stream_t stream);
/* packbits with the threshold min(stats[0]/stats[1], density_threshold) read on device
 * (networks.py:266-268 without the .item() host sync). */
int ngp_packbits_auto(const float* density_grid, int n_bytes, const float* stats,
                      float density_threshold, uint8_t* density_bitfield, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * vren: ray marching            (reference: models/csrc/raymarching.cu:163-454)
 * ------------------------------------------------------------------------------------------ */

/* vren.raymarching_train (binding.cpp:60-81, raymarching.cu:166-332) as a two-call pattern
 * that packs samples in RAY ORDER (deterministic; the reference order comes from atomics):
 *
 *  1. ngp_raymarching_train_count: one march per ray.  Writes rays_a (R,3) i64 =
 *     [ray_idx, start_idx, N_samples] with start_idx the exclusive prefix sum of N_samples in
 *     ray order, counter[0] = S, counter[1] = R (i32), and each ray's sample t values into
 *     t_scratch[r*max_samples + k] (caller provides R*max_samples floats; not zero-filled).
 *  2. the caller reads counter[0] (the only host sync) and allocates S-sized outputs
 *     (counter may be device-mapped pinned host memory: the count then needs no copy);
 *  3. ngp_raymarching_train_write expands the scratch into xyzs,dirs (S,3), deltas,ts (S).
 *
 * hits_t (R,2) f32, noise (R) f32 in [0,1), density_bitfield (cascades*G^3/8) u8. */
int ngp_raymarching_train_count(const float* rays_o, const float* rays_d, const float* hits_t,
                                const uint8_t* density_bitfield, int cascades, float scale,
                                float exp_step_factor, const float* noise, int grid_size,
                                int max_samples, int n_rays,
                                int64_t* rays_a, int32_t* counter, float* t_scratch,
                                ngp_stream_t stream);

int ngp_raymarching_train_write(const float* rays_o, const float* rays_d, const int64_t* rays_a,
                                const float* t_scratch, float scale, float exp_step_factor,
                                int grid_size, int max_samples, int n_rays,
                                float* xyzs, float* dirs, float* deltas, float* ts,
                                ngp_stream_t stream);



/* vren.raymarching_test (binding.cpp:84-106, raymarching.cu:335-454).  hits_t (R_total,2) is
 * advanced in place; alive_indices (N_alive) i64; outputs are dense (N_alive,N_samples,.) and
 * fully written (unused slots zero), N_eff_samples (N_alive) i32.  Keeps the reference's
 * calc_dt(..., cascades) quirk (raymarching.cu:370,399). */
int ngp_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t,
                         const int64_t* alive_indices, const uint8_t* density_bitfield,
                         int cascades, float scale, float exp_step_factor, int grid_size,
                         int max_samples, int n_samples, int n_alive,
                         float* xyzs, float* dirs, float* deltas, float* ts,
                         int32_t* n_eff_samples, ngp_stream_t stream);

/* `alive_indices = alive_indices[alive_indices >= 0]` of the test-time loop (rendering.py:105)
 * on device: survivors of alive_in (n) are appended to alive_out (order not preserved), their
 * number ADDED to count[0] (device i32, caller zeroes); if total_samples is given, the sum of
 * n_eff (n) is ADDED to total_samples[0] (device i64; rendering.py:88). */
int ngp_compact_alive(const int64_t* alive_in, const int32_t* n_eff, int n, int64_t* alive_out,
                      int32_t* count, int64_t* total_samples, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * vren: volume rendering        (reference: models/csrc/volumerendering.cu)
 * ------------------------------------------------------------------------------------------ */

/* vren.composite_train_fw (binding.cpp:109-126, volumerendering.cu:6-84).
 * sigmas,deltas,ts (S) f32; rgbs (S,3) f32; rays_a (R,3) i64.
 * out (all fully written): total_samples (R) i64, opacity,depth (R), rgb (R,3), ws (S). */
int ngp_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas,
                           const float* ts, const int64_t* rays_a, float T_threshold,
                           int n_rays, int n_samples,
                           int64_t* total_samples, float* opacity, float* depth, float* rgb,
                           float* ws, int32_t* n_active_per_ray /* optional (R) i32: min(N, total+1) per row */,
                           ngp_stream_t stream);


/* ngp_composite_train_fw + ngp_active_scan + ngp_nerf_loss for the training step in two launches
 * instead of three (train.py:159-176: render -> NeRFLoss -> mean): every wave also forms its
 * ray's loss terms and backward seeds (dL_drgb (R,3), dL_dopacity (R), already multiplied by
 * grad_scale); one small kernel behind it turns the per-row live-sample counts into ray_offsets
 * (exclusive, row order, total in *n_active) and writes *loss and *sq_err (may be NULL), summed
 * in row order (deterministic).  rays_a must hold exactly one row per ray, as
 * ngp_raymarching_train_count writes it.  bg: 3 floats or NULL.  ray_offsets and workspace
 * (ngp_composite_train_fw_loss_workspace_bytes bytes, scratch) must be 16-byte aligned. */
size_t ngp_composite_train_fw_loss_workspace_bytes(int n_rays);
int ngp_composite_train_fw_loss(const float* sigmas, const float* rgbs, const float* deltas,
                                const float* ts, const int64_t* rays_a, float T_threshold,
                                int n_rays, int n_samples, int64_t* total_samples, float* opacity,
                                float* depth, float* rgb, float* ws, int32_t* ray_offsets,
                                int32_t* n_active, const float* gt_rgb, const float* bg,
                                float lambda_opacity, float grad_scale, float* loss, float* sq_err,
                                float* dL_drgb, float* dL_dopacity, void* workspace,
                                size_t workspace_bytes, ngp_stream_t stream);

/* vren.composite_train_bw (binding.cpp:129-163, volumerendering.cu:87-202).
 * dL_dws may be NULL (treated as zeros).  out: dL_dsigmas (S), dL_drgbs (S,3), fully written. */
int ngp_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                           const float* dL_dws, const float* sigmas, const float* rgbs,
                           const float* ws, const float* deltas, const float* ts,
                           const int64_t* rays_a, const float* opacity, const float* depth,
                           const float* rgb, float T_threshold, int n_rays, int n_samples,
                           float* dL_dsigmas, float* dL_drgbs,
                           const int32_t* ray_offsets, int32_t* active_idx /* both optional: also list the
                           live samples of row n at active_idx[ray_offsets[n] ...] (see ngp_active_scan) */,
                           const float* xyzs, float* x_active /* both optional, with active_idx: also copy the
                           listed samples' positions (S,3) to x_active in list order */,
                           ngp_stream_t stream);

/* vren.composite_test_fw (binding.cpp:166-194, volumerendering.cu:205-285).
 * sigmas,deltas,ts (N_alive,N_samples); rgbs (N_alive,N_samples,3); alive_indices, opacity,
 * depth, rgb are updated in place. */
int ngp_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas,
                          const float* ts, int64_t* alive_indices, float T_threshold,
                          const int32_t* n_eff_samples, int n_alive, int n_samples,
                          float* opacity, float* depth, float* rgb, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * vren: distortion loss         (reference: models/csrc/losses.cu)
 * ------------------------------------------------------------------------------------------ */

/* vren.distortion_loss_fw (binding.cpp:197-209, losses.cu:9-109).
 * out: loss (R) [indexed by ray_idx], ws_inclusive_scan, wts_inclusive_scan (S). */
int ngp_distortion_loss_fw(const float* ws, const float* deltas, const float* ts,
                           const int64_t* rays_a, int n_rays, int n_samples,
                           float* loss, float* ws_inclusive_scan, float* wts_inclusive_scan,
                           ngp_stream_t stream);
/* vren.distortion_loss_bw (binding.cpp:212-231, losses.cu:112-175). out: dL_dws (S). */
int ngp_distortion_loss_bw(const float* dL_dloss, const float* ws_inclusive_scan,
                           const float* wts_inclusive_scan, const float* ws, const float* deltas,
                           const float* ts, const int64_t* rays_a, int n_rays, int n_samples,
                           float* dL_dws, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * tinycudann: multiresolution hash grid   (call site networks.py:36-48; algorithm:
 * NVlabs/tiny-cuda-nn include/tiny-cuda-nn/encodings/grid.h, version unpinned by the reference)
 * ------------------------------------------------------------------------------------------ */

typedef struct ngp_grid_meta {
    int32_t  n_levels;                       /* L (<= NGP_MAX_LEVELS) */
    int32_t  n_features;                     /* F, must be 2 */
    uint32_t offset[NGP_MAX_LEVELS + 1];     /* entry offset of each level; offset[L] = total */
    uint32_t resolution[NGP_MAX_LEVELS];     /* ceil(scale)+1 */
    float    scale[NGP_MAX_LEVELS];          /* exp2(l*log2(b))*N_min - 1 */
} ngp_grid_meta;

/* Fill meta exactly as tiny-cuda-nn does (grid.h GridEncodingTemplated ctor): host side. */
int ngp_grid_meta_init(ngp_grid_meta* meta, int n_levels, int n_features, int log2_hashmap_size,
                       int base_resolution, float per_level_scale);

/* Encode forward.  x (S,3) f32 world positions; x01 = (x - xyz_min)/(xyz_max - xyz_min) is
 * applied in-kernel (networks.py:103; pass min 0 / max 1 for already normalised input).
 * table: (total_entries, 2) f16.  feats: LEVEL-MAJOR [L][S] half2. */
int ngp_hashgrid_fwd(const float* x, const float* xyz_min, const float* xyz_max,
                     const ngp_half* table, const ngp_grid_meta* meta, int n_samples,
                     ngp_half* feats, ngp_stream_t stream);
/* Same with a DEVICE-side sample count (sync-free callers): the launch covers n_samples_max,
 * n_dev[0] (i32, <= n_samples_max) is the real count and the level stride of feats. */
int ngp_hashgrid_fwd_n(const float* x, const float* xyz_min, const float* xyz_max,
                       const ngp_half* table, const ngp_grid_meta* meta, int n_samples_max,
                       const int32_t* n_dev, ngp_half* feats, ngp_stream_t stream);


/* Encode backward w.r.t. the INPUT positions (pose optimisation, train.py:86-89; tiny-cuda-nn computes it
 * when the encoding input requires grad): dL_dx (S,3) f32 = out_scale * sum over levels/features of
 * dfeats * d feat / d x, linear interpolation, table values as stored (f16). */
int ngp_hashgrid_bwd_input(const float* x, const float* xyz_min, const float* xyz_max,
                           const ngp_half* table, const ngp_half* dfeats /* [L][S] half2 */,
                           const ngp_grid_meta* meta, int n_samples, float out_scale,
                           float* dL_dx, ngp_stream_t stream);

/* The same gradient without global atomics (the fast path; DESIGN.md "hash grid backward"):
 * every workgroup owns a <=27600-entry slice of the table in LDS, scans the samples of its level
 * and keeps the updates that fall in its slice (ds_pk_add_f16), then stores the slice.
 * grad_table (total,2) f16 is OVERWRITTEN entirely (no zero-fill needed, no accumulation).
 * active_idx / n_active (both NULL, or both set): compacted backward -- column j of dfeats
 * belongs to sample active_idx[j], j < *n_active (device i32); see ngp_active_samples. */
int ngp_hashgrid_bwd_sliced(const float* x, const float* xyz_min, const float* xyz_max,
                            const ngp_half* dfeats /* [L][S] half2 */, const ngp_grid_meta* meta,
                            int n_samples, const int32_t* active_idx, const int32_t* n_active,
                            ngp_half* grad_table, ngp_stream_t stream);

/* ngp_hashgrid_bwd_sliced with a binning pre-pass: one cheap pass per hashed level writes, per
 * slice, the list of samples whose corners touch it (a sample touches ~4 of a level's 19 slices),
 * and every slice owner then walks its own list instead of all samples.  Same result (f16
 * accumulation order aside), same arguments plus a scratch of
 * ngp_hashgrid_bwd_binned_workspace_bytes(meta, n_samples) bytes. */
size_t ngp_hashgrid_bwd_binned_workspace_bytes(const ngp_grid_meta* meta, int n_samples);
/* active_idx NULL with n_active given: x is already in compact order too (the composite backward's x_active). */
int ngp_hashgrid_bwd_binned(const float* x, const float* xyz_min, const float* xyz_max,
                            const ngp_half* dfeats, const ngp_grid_meta* meta, int n_samples,
                            const int32_t* active_idx, const int32_t* n_active,
                            void* workspace, size_t workspace_bytes,
                            ngp_half* grad_table, ngp_stream_t stream);

/* The same backward in n_groups (<= 16) launches that each COMPLETE one contiguous range of table entries (group 0: the
 * coarse dense levels + the first hashed levels, binning pass included; the other groups: the remaining hashed levels, evenly):
 * a multi-GPU caller hands the finished range [entry_begin, entry_end) x 2 features of grad_table to its gradient
 * collective while the next group's kernel runs (ngp_pl_amd/ddp.py; the reference's DDP buckets, train.py:270-272).
 * Groups must be launched in order 0 .. n_groups-1 on one stream; n_groups = 1 is ngp_hashgrid_bwd_binned. */
int ngp_hashgrid_bwd_binned_group(const float* x, const float* xyz_min, const float* xyz_max,
                                  const ngp_half* dfeats, const ngp_grid_meta* meta, int n_samples,
                                  const int32_t* active_idx, const int32_t* n_active,
                                  void* workspace, size_t workspace_bytes, ngp_half* grad_table,
                                  int n_groups, int group, ngp_stream_t stream);
int ngp_hashgrid_bwd_binned_group_entries(const ngp_grid_meta* meta, int n_samples, int n_groups, int group,
                                          int64_t* entry_begin, int64_t* entry_end);
/* The fused form used by the trainer: ngp_composite_train_fw emits n_active_per_ray, this call
 * turns it IN PLACE into exclusive offsets (+ the total in n_active), and ngp_composite_train_bw
 * writes the list while it walks the rays anyway. */
int ngp_active_scan(int32_t* n_active_per_ray, int n_rays, int32_t* n_active, ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * tinycudann: FullyFusedMLP + SphericalHarmonics  (call sites networks.py:49-77)
 * ------------------------------------------------------------------------------------------ */

/* Weight blob layout (f16, tiny-cuda-nn order: layers in order, each (out,in) row-major,
 * output layer padded to 16 rows):
 *   density net: W0 (64,32) | W1 (16,64)                       = 3072
 *   rgb net:     W0 (64,32) | W1 (64,64) | W2 (16,64)          = 7168
 */
#define NGP_DENSITY_NET_PARAMS 3072
#define NGP_RGB_NET_PARAMS     7168

/* The two halves of NGP.forward (networks.py:94-107,132-153, rgb_act == "Sigmoid"):
 *   density: h = density_net(feats) [f16 (S,16)], sigma = exp(h[0])      (TruncExp forward)
 *   rgb:     sh = SH4(d/|d|), rgb = sigmoid(rgb_net([sh, h]))
 * feats [L=16][S] half2; dirs (S,3) f32 un-normalised.
 * out: sigmas (S) f32; h_out (S,16) f16 (may be NULL for ngp_density_fwd);
 *      rgbs (S,3) f32 (values rounded through f16 as tiny-cuda-nn emits them). */
int ngp_density_fwd(const ngp_half* feats, const ngp_half* density_w, int n_samples,
                    float* sigmas, ngp_half* h_out, ngp_stream_t stream);
/* both in ONE kernel: h stays in registers between the two nets and is only stored when h_out
 * is given (training needs it for the backward; inference passes NULL) */
int ngp_field_fwd(const ngp_half* feats, const float* dirs,
                  const ngp_half* density_w, const ngp_half* rgb_w, int n_samples,
                  float* sigmas, float* rgbs, ngp_half* h_out, ngp_stream_t stream);


/* Backward of the two halves.  Each recomputes its forward, runs dgrad in registers and emits
 * per-workgroup partial weight gradients (n_partials, n_params) f32, n_partials =
 * ngp_field_bwd_partials(n_samples); sum them with ngp_reduce_partials.  All gradients carry
 * the factor loss_scale (tiny-cuda-nn uses 128 for f16).
 *   (colour net, inside ngp_field_bwd): dL_drgbs (S,3) f32 unscaled -> dL_dh (S,16) f16, partials (.,7168)
 *   ngp_density_bwd: dL_dh (S,16) f16 already scaled (may be NULL) and dL_dsigmas (S) f32
 *                    unscaled (may be NULL; TruncExp backward custom_functions.py:168-173 is
 *                    applied here) -> dfeats [L][S] half2, partials (.,3072)
 *   ngp_field_bwd:   both, in ONE launch since round 6: the density net's forward is recomputed from feats and dL_dh
 *                    passes from the colour net to the density net in registers, rounded to f16 as the (S,16)
 *                    hand-over rounded it (dfeats bit-identical to the two-launch form).
 *                    wgrad_partial = [n_partials x 3072 | n_partials x 7168].  h (the forward's h_out) and
 *                    dh_scratch ((S,16) f16 workspace) are no longer read or written and may be NULL; a caller
 *                    that only kept h_out for this call can pass NULL to ngp_field_fwd as well.
 * active_idx / n_active (both NULL or both set) compact the backward as in
 * ngp_hashgrid_bwd_sliced: inputs and f32 seeds are addressed by sample id, dL_dh / dfeats by
 * compact position.  wgrad_partial must be 16-byte aligned (NGP_EINVAL otherwise; also ngp_mlp_bwd). */
int ngp_field_bwd_partials(int n_samples);
int ngp_density_bwd(const ngp_half* feats, const ngp_half* density_w, const ngp_half* dL_dh,
                    const float* dL_dsigmas, float loss_scale, int n_samples,
                    const int32_t* active_idx, const int32_t* n_active,
                    ngp_half* dfeats, float* wgrad_partial, ngp_stream_t stream);
int ngp_field_bwd(const ngp_half* feats, const float* dirs, const ngp_half* h,
                  const ngp_half* density_w, const ngp_half* rgb_w,
                  const float* dL_dsigmas, const float* dL_drgbs, float loss_scale,
                  int n_samples, const int32_t* active_idx, const int32_t* n_active,
                  ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial,
                  ngp_stream_t stream);

/* Generic tcnn.Network / the MLP half of NetworkWithInputEncoding:
 * n_in in {16,32,64} (multiple of 16), 64 neurons, n_hidden in {1,2}, n_out <= 16,
 * out_act 0 None / 1 Sigmoid.  in (S,n_in) f16 row-major, out (S,n_out) f16. */
int ngp_mlp_fwd(const ngp_half* in, const ngp_half* weights, int n_in, int n_hidden, int n_out,
                int out_act, int n_samples, ngp_half* out, ngp_stream_t stream);
int ngp_mlp_bwd_partials(int n_samples);
int ngp_mlp_bwd(const ngp_half* in, const ngp_half* weights, const ngp_half* dL_dout,
                int n_in, int n_hidden, int n_out, int out_act, int n_samples,
                ngp_half* dL_din /* may be NULL */, float* wgrad_partial, ngp_stream_t stream);

/* tcnn.Encoding SphericalHarmonics degree 4 (networks.py:58-65): in (S,3) f32 in [0,1]
 * (the module maps it back to [-1,1]), out (S,16) f16. */
int ngp_sh4_fwd(const float* dirs01, int n_samples, ngp_half* out, ngp_stream_t stream);
/* its backward w.r.t. the input: dL_ddirs01 (S,3) f32 = out_scale * 2 * dSH/d(xyz) . dL_dsh (S,16) f16 */
int ngp_sh4_bwd(const float* dirs01, const ngp_half* dL_dsh, int n_samples, float out_scale,
                float* dL_ddirs01, ngp_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * optimizer: apex FusedAdam equivalent (train.py:131, eps 1e-15) fused with AMP plumbing
 * ------------------------------------------------------------------------------------------ */

/* One dense pass over n params:  g = grad/grad_scale (grad f16 or f32), Adam update of the f32
 * master `param` with moments m,v, write the f16 working copy `param_h` (may be NULL), and zero
 * the gradient.  step is 1-based.  If found_inf (device i32, may be NULL) is non-zero the update
 * is skipped but gradients are still zeroed (GradScaler semantics). */
int ngp_adam_step(float* param, ngp_half* param_h, void* grad, int grad_is_f32,
                  float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, const int32_t* found_inf,
                  ngp_stream_t stream);
/* The optimizer step of the whole field in ONE launch (train.py:131 hands every parameter of the
 * model to one FusedAdam): ngp_adam_step (f16 gradient) on the grid table and
 * ngp_adam_step_partials on the density and rgb MLP blocks, bit-identical to the three separate
 * calls.  The MLP workgroups are dispatched first and run underneath the HBM-bound grid pass.
 * zero_grid_grad = 0 leaves grid_grad as it is (28 instead of 30 bytes per parameter): the sliced / binned table
 * backwards OVERWRITE the whole gradient table every step, only accumulating producers (ngp_hashgrid_bwd) need it cleared. */
int ngp_adam_step_field(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad,
                        float* grid_m, float* grid_v, int64_t n_grid,
                        float* density_param, ngp_half* density_param_h,
                        const float* density_partials, float* density_m, float* density_v,
                        int n_density,
                        float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                        float* rgb_m, float* rgb_v, int n_rgb,
                        int n_partials, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int zero_grid_grad,
                        const int32_t* found_inf, int32_t* step_state, ngp_stream_t stream);
/* step_state (may be NULL: `step` is the bias-correction step, as everywhere else): 4 x i32 on the device holding the number of
 * APPLIED steps -- {MLP blocks: slot 0, slot 1; grid block: slot 0, slot 1}, zeroed (or set to the steps already taken) by the
 * caller once.  With it `step` is the 1-based number of this CALL: the launch reads slot (step - 1) & 1, corrects the bias for
 * applied + 1, and writes slot step & 1 = applied + (skipped by its found_inf flag ? 0 : 1).  apex / GradScaler leave the
 * optimizer's step count unchanged on a skipped step; the flag lives on the device, so the count has to as well. */
/* The same launch for a data-parallel rank that owns ONE SHARD of the grid table (ngp_pl_amd/ddp.py ShardedExchange: reduce-scatter of
 * the gradient -> this update -> all-gather of the updated f16 table): the grid pointers address the rank's shard (n_shard
 * parameters, gradient = the reduce-scatter's output), the MLP blocks are updated by every rank alike.  Two skip flags: the MLP
 * blocks follow found_inf_mlp (computed on the all-reduced MLP sums: identical on every rank), the shard follows found_inf_shard
 * (computed on the shard's reduced gradient by its owner): every parameter is decided by exactly one flag that all ranks that
 * update it agree on, so the ranks stay in lock step without a flag collective.  The gradient is not cleared.  n_shard may be 0
 * (a rank whose shard is empty: the MLP blocks only).  step_state: as above, one count per flag. */
int ngp_adam_step_field_shard(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad,
                              float* grid_m, float* grid_v, int64_t n_shard,
                              float* density_param, ngp_half* density_param_h,
                              const float* density_partials, float* density_m, float* density_v,
                              int n_density,
                              float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                              float* rgb_m, float* rgb_v, int n_rgb,
                              int n_partials, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale,
                              const int32_t* found_inf_mlp, const int32_t* found_inf_shard, int32_t* step_state,
                              ngp_stream_t stream);
/* The same launch for the CHUNKED exchange of the native data-parallel step (ngp_stepper_tail below): the table gradient is
 * reduce-scattered in n_chunks chunks of world x piece f16 values (piece a multiple of 8; n_chunks x world x piece >= n_grid, the
 * buffers are padded to that); this rank owns values [c * world * piece + rank * piece, + piece) of every chunk c.  grid_* address
 * the WHOLE table (n_grid parameters), shard_grad holds the rank's n_chunks pieces back to back (the reduce-scatters' outputs). */
int ngp_adam_step_field_pieces(float* grid_param, ngp_half* grid_param_h, const ngp_half* shard_grad,
                               float* grid_m, float* grid_v, int64_t n_grid, int64_t piece,
                               int32_t n_chunks, int32_t world, int32_t rank,
                               float* density_param, ngp_half* density_param_h,
                               const float* density_partials, float* density_m, float* density_v,
                               int n_density,
                               float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                               float* rgb_m, float* rgb_v, int n_rgb,
                               int n_partials, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float grad_scale,
                               const int32_t* found_inf_mlp, const int32_t* found_inf_shard, int32_t* step_state,
                               ngp_stream_t stream);
/* GradScaler's non-finite check (train.py:274 precision=16 -> torch.amp.GradScaler.unscale_) on a native
 * gradient buffer of n elements (f16, or f32 if grad_is_f32; 16-byte aligned): flag[0] (device i32) |= 1 if any
 * element is inf or NaN; reset != 0 zeroes the flag first.  Used behind the multi-GPU all-reduce, whose f16
 * sum can overflow where no single rank's gradient did; feed the flag to ngp_adam_step*'s found_inf. */
int ngp_found_inf(const void* grad, int grad_is_f32, int64_t n, int32_t* flag, int reset,
                  ngp_stream_t stream);
/* The same check over two buffers in one launch (byte sizes multiples of 16); flag_clear (may be NULL) is zeroed on the side:
 * a caller alternating between two flags needs no memset launch per step. */
int ngp_found_inf2(const void* grad_a, int a_is_f32, int64_t n_a, const void* grad_b, int b_is_f32, int64_t n_b,
                   int32_t* flag, int32_t* flag_clear, ngp_stream_t stream);
/* Sum n_partials rows of (n) f32 into out (n) f32 (out = sum, not accumulated). */
int ngp_reduce_partials(const float* partials, int n_partials, int n, float* out,
                        ngp_stream_t stream);
/* Both MLP blocks in one launch: out[0:n_a] and out[n_a:n_a+n_b] = column sums of partials_a (n_partials, n_a) and
 * partials_b (n_partials, n_b). */
int ngp_reduce_partials2(const float* partials_a, int n_a, const float* partials_b, int n_b,
                         int n_partials, float* out, ngp_stream_t stream);
/* f32 -> f16 cast (tiny-cuda-nn casts the master params every forward). */
int ngp_cast_f32_to_f16(const float* in, int64_t n, ngp_half* out, ngp_stream_t stream);
int ngp_cast_f16_to_f32(const ngp_half* in, int64_t n, float scale, float* out,
                        ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * loss seeds (losses.py:47-60 + the mean reduction of train.py:173)
 * ------------------------------------------------------------------------------------------ */

/* With rgb_f = rgb + bg*(1-opacity) (bg (3) f32 or NULL = black; rendering.py:153-161):
 *   loss[0] = mean((rgb_f-gt)^2) + mean(lambda_o * -(o+1e-10) log(o+1e-10))
 * and the backward seeds dL_drgb (R,3), dL_dopacity (R) w.r.t. the COMPOSITED rgb/opacity, both
 * multiplied by grad_scale.  loss (1) f32 and sq_err (1) f32 (may be NULL; sum((rgb_f-gt)^2), for
 * PSNR) are OVERWRITTEN.  Batches of up to 16384 rays reduce through a library-owned scratch
 * (deterministic, no atomics): do not run two of these launches concurrently on different streams. */
int ngp_nerf_loss(const float* rgb, const float* opacity, const float* gt_rgb, const float* bg,
                  float lambda_opacity, float grad_scale, int n_rays,
                  float* loss, float* sq_err, float* dL_drgb, float* dL_dopacity,
                  ngp_stream_t stream);

/* NeRFLoss.forward as the reference shapes it (losses.py:47-60): UNREDUCED terms
 *   sq_err (R,3) = (rgb - gt)^2,   entropy (R) = lambda_o * -(o + 1e-10) log(o + 1e-10)
 * (train.py:173 sums their means) and the backward through them: g_rgb = g_sq_err * 2 (rgb - gt),
 * g_opacity = g_entropy * lambda_o * -(log(o + 1e-10) + 1).  One launch each.
 * `*_is_scalar`: the seed is ONE float broadcast over all elements (what `.mean().backward()` hands back as a stride-0 view). */
int ngp_nerf_loss_terms_fw(const float* rgb, const float* opacity, const float* gt_rgb,
                           float lambda_opacity, int n_rays, float* sq_err, float* entropy,
                           ngp_stream_t stream);
int ngp_nerf_loss_terms_bw(const float* g_sq_err, int g_sq_err_is_scalar, const float* g_entropy,
                           int g_entropy_is_scalar, const float* rgb, const float* opacity,
                           const float* gt_rgb, float lambda_opacity, int n_rays, float* g_rgb,
                           float* g_opacity, ngp_stream_t stream);
/* render()'s background blend rgb_out = rgb + bg (1 - opacity) (rendering.py:153-161; bg 3 floats on the device) and its backward
 * onto the opacity seed: g_opacity_out = g_opacity (NULL: 0) - sum_c g_rgb[.,c] bg[c]. */
int ngp_bg_blend(const float* rgb, const float* opacity, const float* bg, int n_rays, float* rgb_out, ngp_stream_t stream);
int ngp_bg_blend_bw(const float* g_rgb, const float* g_opacity, const float* bg, int n_rays, float* g_opacity_out,
                    ngp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * batch sampling (datasets/base.py:22-35 'all_images' + train.py:78-91 + ray_utils.py:46-70)
 * ------------------------------------------------------------------------------------------ */

/* Draws n (image, pixel) pairs uniformly (counter-based RNG keyed by seed), gathers the pixel
 * colour and forms the world-space ray: rays_d = R * direction (not normalised), rays_o = camera
 * centre.  poses (n_images,3,4) f32 c2w; directions (n_pixels,3) f32; images (n_images,n_pixels,3)
 * f32.  Optional outputs: noise (n) f32 in [0,1) for the marcher's jitter; img_idx, pix_idx (n) i32. */
int ngp_sample_rays(const float* poses, const float* directions, const float* images,
                    int n_images, int n_pixels, int n, uint64_t seed,
                    float* rays_o, float* rays_d, float* rgb, float* noise,
                    int32_t* img_idx, int32_t* pix_idx, ngp_stream_t stream);

/* datasets/ray_utils.py:50-74 get_rays for one camera: rays_d (n,3) = directions (n,3) @ c2w[:, :3].T (not normalised),
 * rays_o (n,3) = the camera centre c2w[:, 3] repeated.  c2w (3,4) f32 row-major ON THE DEVICE.  (The reference's FPS figure
 * times ray generation together with render(): test.ipynb cell 2.) */
int ngp_get_rays(const float* directions, const float* c2w, int n, float* rays_o, float* rays_d, ngp_stream_t stream);

/* ---- occupancy-grid maintenance -------------------------------------------------------- */


/* NGP.mark_invisible_cells (networks.py:197-238; train.py:155-158 runs it once before training): for every cell of every
 * cascade, count_grid (C, G^3) f32 = the fraction of the n_cams training cameras that have the cell centre inside their image at
 * depth >= near_distance, density_grid (C, G^3) f32 = 0 where that fraction is > 0 and no camera has the centre inside its image
 * closer than near_distance, else -1 (such cells are never marched nor updated).  K (3,3) f32 intrinsics, poses (n_cams,3,4) f32
 * camera-to-world, both on the device; grids in Morton order.  Any n_cams >= 1 (the cameras pass through LDS 1024 at a time). */
int ngp_mark_invisible_cells(const float* K, const float* poses, int n_cams, int img_w, int img_h, float near_distance,
                             int cascades, int grid_size, float scale, float* count_grid, float* density_grid,
                             ngp_stream_t stream);

/* NGP.update_density_grid (networks.py:240-269 with get_all_cells :155-167 and
 * sample_uniform_and_occupied_cells :169-195) in one call, no host sync:
 *   per cascade: warmup ? every cell : G^3/4 uniform cells + G^3/4 cells uniform over
 *   {density_grid > density_threshold} (inverse-CDF lookup instead of nonzero()+randint());
 *   sigma at the jittered cell centres (hash grid + density MLP); then
 *   grid = (grid < 0) ? grid : max(grid * decay, sigma or 0), threshold = min(mean(grid > 0),
 *   density_threshold), pack bits.
 * density_grid (C, G^3) f32 Morton order, density_bitfield (C*G^3/8) u8, both updated in place;
 * decay_grid (C, G^3) f32 per-cell decay or NULL (erode, networks.py:262-264); seed keys the
 * counter-based RNG (pass the step number).  workspace: ngp_occupancy_update_workspace_bytes. */
size_t ngp_occupancy_update_workspace_bytes(int cascades, int grid_size);
int ngp_occupancy_update(float* density_grid, uint8_t* density_bitfield, int cascades, int grid_size,
                         float scale, float density_threshold, float decay, const float* decay_grid,
                         int warmup, uint64_t seed,
                         const float* xyz_min, const float* xyz_max, const ngp_half* table,
                         const ngp_grid_meta* meta, const ngp_half* density_w,
                         void* workspace, size_t workspace_bytes, ngp_stream_t stream);
/* ---- test-time frame loop -------------------------------------------------------------- */

/* The whole test-time loop `__render_rays_test` (rendering.py:46-118) for one batch of rays,
 * device-driven: the alive-ray count, N_samples (rendering.py:69-70) and the batch sizes live in
 * device memory, every kernel of an iteration is launched for an upper bound that the host learns
 * two iterations late, so the GPU never waits for the host.  Field = hash grid + the two MLPs as
 * in ngp_hashgrid_fwd / ngp_field_fwd.
 *   hits_t (R,2) f32: (near, far) per ray after the NEAR_DISTANCE clamp (rendering.py:27-29);
 *     read only.  min_samples = exp_step_factor == 0 ? 1 : 4 (rendering.py:60).
 *   chunk_scale >= 1 multiplies the reference's N_samples = N_rays // N_alive (fewer, larger
 *     iterations); probe_cap > 0 bounds the grid probes of one ray in one iteration and retires a
 *     ray as soon as it reaches its far hit.  chunk_scale == 1 && probe_cap == 0 reproduces the
 *     reference's chunking exactly (bit-identical to the loop over ngp_raymarching_test /
 *     ngp_composite_test_fw); other settings emit the SAME samples per ray and differ only in
 *     where `T = 1 - opacity` is re-read between chunks (volumerendering.cu:229), i.e. by float
 *     rounding of the composite.
 *   bg: HOST pointer to 3 floats, rgb += bg * (1 - opacity) at the end (rendering.py:112-116);
 *     NULL = no blend.
 *   workspace: device scratch of ngp_render_test_workspace_bytes(n_rays, chunk_scale,
 *     exp_step_factor) bytes.
 *   out: opacity, depth (R) f32, rgb (R,3) f32, total_samples (1) i64 on device (sum of
 *     N_eff_samples, rendering.py:88); n_iterations (HOST i32, may be NULL).
 * Returns after all work is enqueued on `stream` (it waits on its own lagged events only). */
size_t ngp_render_test_workspace_bytes(int n_rays, int chunk_scale, float exp_step_factor);
int ngp_render_test_frame(const float* rays_o, const float* rays_d, const float* hits_t,
                          const uint8_t* density_bitfield, int cascades, float scale,
                          float exp_step_factor, int grid_size, int max_samples, float T_threshold,
                          const float* xyz_min, const float* xyz_max, const ngp_half* table,
                          const ngp_grid_meta* meta, const ngp_half* density_w, const ngp_half* rgb_w,
                          int n_rays, int chunk_scale, int probe_cap, const float* bg,
                          void* workspace, size_t workspace_bytes,
                          float* opacity, float* depth, float* rgb, int64_t* total_samples,
                          int32_t* n_iterations, ngp_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * native driver of one optimisation step   (reference: NeRFSystem.training_step, train.py:159-185, with
 * configure_optimizers :112-139; the kernels are the entry points above)
 * ------------------------------------------------------------------------------------------
 * One object per model and batch size.  The caller owns every buffer (device pointers below; nothing here allocates device
 * memory) and keeps them alive while the stepper exists.  A step is three calls, so that a multi-GPU caller can place its
 * gradient collectives between them (ngp_pl_amd/ddp.py):
 *     ngp_stepper_front          march hand-over -> sample expansion -> hash grid -> field -> composite + loss seeds ->
 *                                composite backward -> field backward (per-workgroup weight-gradient partials)
 *     ngp_stepper_table_backward the hash-table gradient (binned; one-pass kernel for batches beyond the bin workspace)
 *     ngp_stepper_update         fused Adam on the native gradient buffers
 * and ngp_stepper_march enqueues AABB + jitter + pass 1 of the march of a batch on the marching stream (front() does it
 * for the NEXT batch behind the composite forward when it is handed one: the march then overlaps this step's backward).
 * The only host wait of a step is front()'s poll of the pinned sample count of its own batch's march, bounded by
 * NGP_SPIN_TIMEOUT_S (NGP_ETIMEOUT).  The occupancy-grid update stays with the caller (ngp_occupancy_update between
 * update() and march() every `update_interval` steps, train.py:160-163). */
typedef struct ngp_stepper_config {
    /* scene + occupancy grid */
    const float* center; const float* half_size; const float* xyz_min; const float* xyz_max;
    const uint8_t* density_bitfield;
    int32_t cascades, grid_size;
    float scale, exp_step_factor;
    ngp_grid_meta meta;
    /* parameters: f32 masters, f16 working copies, Adam moments.  enc = [density MLP (n_density) | grid table (n_grid)] */
    float* enc_param; ngp_half* enc_half; float* enc_m; float* enc_v;
    float* rgb_param; ngp_half* rgb_half; float* rgb_m; float* rgb_v;
    int64_t n_grid; int32_t n_density, n_rgb;
    ngp_half* grid_grad16;                         /* (n_grid) packed f16, overwritten by every table backward */
    /* recipe */
    int32_t max_samples;                           /* rendering.py:7 MAX_SAMPLES */
    float near_distance, T_threshold, lambda_opacity, lambda_distortion;
    const float* bg;                               /* 3 floats on the device, or NULL (black) */
    float beta1, beta2, eps, weight_decay;
    uint64_t noise_seed;
} ngp_stepper_config;

typedef struct ngp_step_buffers {
    int32_t n_rays; int32_t distortion;            /* distortion != 0: ws_incl, wts_incl, dL_dws, dist, dist_seed are used */
    int64_t cap;                                   /* sample slots of the per-sample buffers (n_rays * max_samples) */
    /* per sample */
    float* xyzs; float* dirs; float* deltas; float* ts; ngp_half* feats; ngp_half* h; float* sigmas; float* rgbs; float* ws;
    float* dL_dsigmas; float* dL_drgbs; int32_t* active; float* x_act; ngp_half* dh; ngp_half* dfeats;
    float* ws_incl; float* wts_incl; float* dL_dws;
    /* per ray */
    int64_t* total; float* opacity; float* depth; float* rgb; float* dL_drgb; float* dL_dopacity; int32_t* ray_offs;
    float* dist; const float* zeros; const float* dist_seed;
    /* two sets of march records (the march of batch k+1 runs while step k reads its own) */
    float* hits_t[2]; int64_t* rays_a[2]; float* noise[2]; float* scratch[2];
    int32_t* counter[2];                           /* pinned, device-mapped host memory, 4 x i32 each: {S, R} of a march, [2] = the live-sample
                                                      count of the step that consumed it (written by the composite's tail kernel) */
    /* two-round forward (optional: all three NULL disables it) */
    int32_t* list_k;                               /* (n_rays * 64) ids of the rays' first samples */
    int32_t* list_rest;                            /* (cap) ids of the continuing rays' remaining samples */
    int32_t* two_round_counts;                     /* 4 x i32 on the device, zeroed by the caller once */
    int32_t* offs_k[2];                            /* (n_rays) per march record set: where each ray's first samples go in the compact
                                                      first-round list (NULL: the padded list is used); counter[.][3] = its length */
    /* scalars and workspaces */
    int32_t* n_active; float* stats;               /* stats[0] = loss, stats[1] = sum of squared errors */
    float* partials; int32_t max_partials;         /* [max_partials x (n_density + n_rgb)] f32 */
    void* fw_ws; size_t fw_bytes;                  /* ngp_composite_train_fw_loss_workspace_bytes(n_rays) */
    void* bin_ws; size_t bin_bytes; int32_t bin_max; /* binned table backward: workspace for up to bin_max samples (0: always one-pass) */
} ngp_step_buffers;

/* Two-round forward (NGP_TWO_ROUND = auto (default) | on | off; first K = NGP_TWO_ROUND_K, default 32).  Late in training a few per
 * cent of the marched samples lie in front of their ray's early stop (measured: 5 % after 25 000 steps), yet hash grid and field
 * were evaluated on all of them.  When the previous step's live fraction is below 0.15 (back above 0.25: off again) front() runs:
 * sample expansion + list of every ray's first K samples (compact, at the offsets the march's scan prepared in offs_k; padded
 * where it did not) -> hash grid + field on that list -> ngp_composite_probe lists
 * the rest of the rays that are still transparent -> hash grid + field on that list -> the unchanged composite.  Exactly
 * equivalent: the composite never reads a sample behind a ray's stop (volumerendering.cu:20-44), and every sample in front of it
 * has been evaluated by the same per-sample kernels (tests/test_train_gpu.py::test_two_round_forward_is_bit_identical).  What it
 * buys is modest because rays stop deep, not early (the occupied shell in front of a surface is ~20 samples thick): 0.304 -> 0.288
 * ms per step after 8 000 steps, 0.302 -> 0.283 after 25 000 (profiles/archive_r01_r04/r03_two_round_forward.txt); K = 8 loses.  Not used with
 * the distortion loss (its kernels walk every sample's ws) and never at the bench's operating point (half of the samples live). */
typedef struct ngp_stepper ngp_stepper;
int ngp_stepper_create(const ngp_stepper_config* config, const ngp_step_buffers* buffers, ngp_stepper** out);
int ngp_stepper_destroy(ngp_stepper* s);
/* New buffers (another batch size).  Any prefetched march is waited for and dropped first. */
int ngp_stepper_set_buffers(ngp_stepper* s, const ngp_step_buffers* buffers);
/* A second set of packed-sample buffers (each sized like its namesake in ngp_step_buffers; all four or all NULL = back to one
 * set).  With two sets a march also EXPANDS its samples (pass 2 of the march, xyzs / dirs / deltas / ts) on the marching stream,
 * into the set that belongs to its record set, while the running step reads the other one: the expansion leaves the main stream
 * (9 us of kernel and a launch gap per step; same kernel, same inputs: bit-identical samples).  The set a step read:
 * ngp_stepper_last_set() (0 = the ngp_step_buffers pointers, 1 = these).  A pending march is waited for and dropped first;
 * ngp_stepper_set_buffers() forgets the second set.  Marches whose scan prepares the two-round lists expand on the main stream
 * as before. */
int ngp_stepper_set_sample_sets(ngp_stepper* s, float* xyzs1, float* dirs1, float* deltas1, float* ts1);
/* AABB + near clamp + jitter + pass 1 of the march of (rays_o, rays_d) on `march_stream`, behind everything `main_stream`
 * has queued so far.  One march can be pending; NGP_EINVAL if one already is. */
int ngp_stepper_march(ngp_stepper* s, const float* rays_o, const float* rays_d, ngp_stream_t main_stream, ngp_stream_t march_stream);
/* Is a march of exactly these buffers pending?  (1 / 0) */
int ngp_stepper_pending(const ngp_stepper* s, const float* rays_o, const float* rays_d);
/* Which of the two march record sets (hits_t / rays_a / noise / scratch / counter) the last front() consumed: 0 or 1. */
int ngp_stepper_last_set(const ngp_stepper* s);
/* sizeof(ngp_stepper_config) (which = 0) / sizeof(ngp_step_buffers) (which = 1) / sizeof(ngp_exchange_config) (which = 2) as this
 * library was compiled: a binding that
 * mirrors the records (ctypes, cgo ...) checks its own layout against them before it hands one over. */
int ngp_stepper_record_bytes(int which);
/* Waits for a pending march and forgets it (the batch it was made for is not going to be stepped). */
int ngp_stepper_drop_pending(ngp_stepper* s);
/* The step up to the field backward, on main_stream.  The pending march must be the one of (rays_o, rays_d).
 * next_o / next_d (may be NULL): the following batch, marched behind the composite forward.  loss_scale: factor on the
 * f16 backward (128, or 128 / world under data parallelism); grad_scale: GradScaler-style factor on the loss seeds.
 * out (host): n_samples = S of this batch, n_partials = rows of weight-gradient partials the field backward wrote. */
int ngp_stepper_front(ngp_stepper* s, const float* rays_o, const float* rays_d, const float* rgb_gt,
                      const float* next_o, const float* next_d, float loss_scale, float grad_scale,
                      ngp_stream_t main_stream, ngp_stream_t march_stream, int32_t* n_samples, int32_t* n_partials);
int ngp_stepper_table_backward(ngp_stepper* s, int n_groups, int group, ngp_stream_t main_stream);
/* table_backward(1, 0) + update() of the single-process step in ONE call, with the dense levels' merge folded into the Adam launch
 * (ngp_hashgrid_bwd_binned_deferred + ngp_adam_step_field_merge: one launch and one pass over 0.5 M entries less on the critical
 * path; bit-identical parameters).
 * OVERFLOW GUARD (what torch.cuda.amp.GradScaler does for the reference under Lightning's precision=16, train.py:274): the field
 * backward of front() raises a device flag when a weight-gradient sum is inf / NaN -- an f16 overflow anywhere in the recomputed
 * forward or the backward chain ends there --, and this call (like ngp_stepper_update without a found_inf of the caller's) hands the
 * flag to the optimizer launch: such a step changes no parameter and no moment.  step_state (may be NULL): the device-side counts of
 * APPLIED steps of ngp_adam_step_field, so that the bias correction does not advance on a skipped step. */
int ngp_stepper_backward_update(ngp_stepper* s, float lr, int32_t step, float grad_scale, int32_t* step_state, ngp_stream_t main_stream);
/* The same step split where the reference's API splits it (render() -> NeRFLoss -> autograd, rendering.py:121-163, losses.py:47-60):
 * render_forward = front() up to the composite WITHOUT the loss (plain ngp_composite_train_fw + ngp_active_scan), the
 * background blend into rgb_out (R,3; may be NULL) and the next batch's march; the per-ray / per-sample results stay in the
 * step buffers (opacity, depth, rgb, ws, total, deltas, ts, rays_a of set ngp_stepper_last_set()).
 * render_backward = the rest of front() from the caller's seeds: g_rgb (R,3) w.r.t. the BLENDED colour, g_opacity, g_depth (R),
 * g_ws (S) w.r.t. the composited values (the last three may be NULL = zero).  A stepper used only this way may be created
 * without f32 masters / Adam moments (NULL): ngp_stepper_update then returns NGP_EINVAL. */
int ngp_stepper_render_forward(ngp_stepper* s, const float* rays_o, const float* rays_d, const float* next_o, const float* next_d,
                               float* rgb_out, ngp_stream_t main_stream, ngp_stream_t march_stream, int32_t* n_samples);
int ngp_stepper_render_backward(ngp_stepper* s, const float* g_rgb, const float* g_opacity, const float* g_depth, const float* g_ws,
                                float loss_scale, ngp_stream_t main_stream, int32_t* n_partials);
/* Fused Adam (ngp_adam_step_field).  density_partials / rgb_partials NULL: the partial rows front() wrote (n_partials rows);
 * a caller that reduced them across ranks passes its own buffers with n_partials = 1.  grad_scale = the total factor the
 * gradients carry (loss_scale x grad_scale x world).  step is 1-based (bias correction; with step_state -- see
 * ngp_adam_step_field -- the number of this call, the applied-step count then lives on the device next to found_inf). */
int ngp_stepper_update(ngp_stepper* s, float lr, int32_t step, float grad_scale, const float* density_partials,
                       const float* rgb_partials, int32_t n_partials, const int32_t* found_inf, int32_t* step_state,
                       ngp_stream_t main_stream);
/* Dynamic loss scale of the native step = torch.cuda.amp.GradScaler on the device (the reference trains under Lightning's precision=16,
 * train.py:274: GradScaler's scale on top of tiny-cuda-nn's fixed 128; without it half of the feature gradients of a step flush to
 * zero in f16 and nearly all others are subnormal -- round 6, profiles/r06_loss_scale.txt).  The field backward multiplies its f32 seeds by
 * loss_scale x scale, the optimizer launch divides by it (powers of two: exact) and applies GradScaler's rule next to the overflow
 * guard's flag: scale x backoff_factor after a skipped step, x growth_factor after growth_interval clean ones (defaults of
 * torch: 65536, 2, 0.5, 2000).  No host sync anywhere.  init_scale <= 0: off (the fixed loss_scale alone, as before).  Synchronises
 * main_stream once (at configuration). */
int ngp_stepper_set_loss_scaler(ngp_stepper* s, float init_scale, float growth_factor, float backoff_factor, int32_t growth_interval,
                                ngp_stream_t main_stream);
/* The scale the NEXT step will use and its count of clean steps (synchronises main_stream; 0 when the scaler is off). */
int ngp_stepper_loss_scale(ngp_stepper* s, float* scale, int32_t* growth_tracker, ngp_stream_t main_stream);
/* Host-side accounting since the last reset: seconds the entry points spent polling for a march's sample count (device-bound
 * wait) and in everything else (argument checks, launches, event records), and the number of front() calls. */
int ngp_stepper_host_times(ngp_stepper* s, double* wait_s, double* enqueue_s, long long* n_steps, int reset);
/* Stage timing (HIP events on the streams the kernels run on).  enable != 0: the following steps record events;
 * ngp_stepper_stage_times synchronises and writes the last step's times in ms:
 *   [0] march_write [1] hashgrid_fwd [2] mlp_fwd [3] composite_fw+loss [4] composite_bw [5] mlp_bwd [6] hashgrid_bwd [7] adam
 *   [8] march_count (marching stream; of the batch the last front() consumed).  Negative = not recorded. */
#define NGP_STEPPER_STAGES 9
int ngp_stepper_timing(ngp_stepper* s, int enable);
int ngp_stepper_stage_times(ngp_stepper* s, float* ms);


/* ------------------------------------------------------------------------------------------
 * data-parallel exchange   (reference: Lightning DDPPlugin, train.py:268-272, opt.py:42: one process per GPU, the gradients of
 * every step averaged over the ranks; here on the native gradient buffers, over RCCL / xGMI)
 * ------------------------------------------------------------------------------------------
 * ngp_comm = an RCCL communicator + its own high-priority HIP stream.  RCCL is resolved at run time (dlopen librccl.so.1): the
 * library does not link against it.  One rank creates the id, the caller carries its 128 bytes to the other ranks (bench.py:
 * a torch.distributed broadcast), every rank calls ngp_comm_create on ITS device (hipSetDevice first).  The collectives below
 * enqueue on `stream` (NULL: the communicator's stream); in-place where there is one buffer.  dtype: NGP_COMM_F32 / NGP_COMM_F16. */
#define NGP_COMM_ID_BYTES 128
#define NGP_COMM_F32 0
#define NGP_COMM_F16 1
typedef struct ngp_comm ngp_comm;
const char* ngp_comm_last_error(void);
int ngp_comm_unique_id(void* id_bytes);
int ngp_comm_create(const void* id_bytes, int world, int rank, ngp_comm** out);
int ngp_comm_destroy(ngp_comm* c);
int ngp_comm_info(const ngp_comm* c, int32_t* world, int32_t* rank, int32_t* rccl_version, ngp_stream_t* stream);
int ngp_comm_all_reduce(ngp_comm* c, void* buf, int64_t count, int dtype, ngp_stream_t stream);
/* recv (recv_count) = this rank's slice of the sum over ranks of send (world x recv_count) */
int ngp_comm_reduce_scatter(ngp_comm* c, const void* send, void* recv, int64_t recv_count, int dtype, ngp_stream_t stream);
/* recv (world x send_count) = the ranks' send buffers in rank order; in place when send == recv + rank x send_count */
int ngp_comm_all_gather(ngp_comm* c, const void* send, void* recv, int64_t send_count, int dtype, ngp_stream_t stream);
int ngp_comm_broadcast(ngp_comm* c, void* buf, int64_t n_bytes, int root, ngp_stream_t stream);
/* Point-to-point forms (xGMI is a full mesh of point-to-point links; a ring collective is bound by one of them).
 * _exchange_slices: slice q of send (world x count) goes straight to rank q, rank q's contribution lands in recv + q x count; the
 *   own slice is not copied.  The data movement of a reduce-scatter without its additions: the caller sums the world slices itself
 *   (ngp_sum_slices_f16: in rank order, in f32 -- deterministic).
 * _all_gather_direct: in place; this rank's slice buf + rank x count goes to every peer, theirs arrive at buf + q x count.
 * Both are ONE RCCL group of world - 1 sends and receives; no-ops at world 1. */
int ngp_comm_exchange_slices(ngp_comm* c, const void* send, void* recv, int64_t count, int dtype, ngp_stream_t stream);
int ngp_comm_all_gather_direct(ngp_comm* c, void* buf, int64_t count, int dtype, ngp_stream_t stream);
/* out[i] (f16) = round(sum over q = 0 .. world-1, in that order, in f32, of (q == rank ? own[i] : stage[q x count + i])). */
int ngp_sum_slices_f16(const ngp_half* own, const ngp_half* stage, int world, int rank, int64_t count, ngp_half* out,
                       ngp_stream_t stream);

/* The tail of a data-parallel step, enqueued natively: with an exchange installed, ngp_stepper_tail replaces
 * ngp_stepper_table_backward + ngp_stepper_update.  Per step, EVERY rank issues the same sequence (a rank whose batch had no
 * samples contributes zeros):
 *   main stream:  MLP partial rows -> sums (`small`)  |  table backward in n_groups launch groups
 *   comm stream:  all-reduce(small)  |  per chunk c, behind the launch group that completes it: reduce-scatter (mode 1) or
 *                 all-reduce (mode 0) of chunk c of the gradient  |  non-finite checks on the REDUCED buffers  |  Adam (mode 1: this
 *                 rank's pieces + the MLP blocks; mode 0: the whole table)  |  mode 1: all-gather of the updated f16 table chunks
 * The main stream only RECORDS events for the communicator's stream and waits once, at the end of the tail (the next step's
 * forward and the occupancy update read the updated table).  The gradients were produced at loss scale 128 / world, so their SUM
 * sits at the single-GPU scale: f16 headroom and underflow floor do not depend on the world size.
 *   mode 1 ("sharded", default): the optimizer pass is divided by the world size and what is gathered is the table the next
 *     forward reads; a rank's f32 master / moments are current inside its own pieces only.
 *   mode 0 ("allreduce"): the reference's semantics literally -- gradient all-reduce only, every rank updates everything.
 *   mode 2 ("direct", round 5; n_chunks must be 1): mode 1 with the two ring collectives replaced by point-to-point transfers over
 *     all xGMI links at once: ngp_comm_exchange_slices moves every peer's slice of this rank's share into `stage`, ngp_sum_slices_f16
 *     adds the world slices in rank order in f32 (deterministic; the ring adds f16 in ring order), Adam runs on the share, and
 *     ngp_comm_all_gather_direct sends the updated f16 share to every peer.
 * Buffers are the caller's: grad_padded / table_padded must be the stepper's grid_grad16 / enc_half + n_density, allocated with
 * n_chunks x world x piece values (padding zero). */
typedef struct ngp_exchange_config {
    int32_t mode, n_chunks, n_groups, reserved;
    int64_t piece;                 /* f16 values per rank and chunk, a multiple of 8 */
    ngp_half* grad_padded; ngp_half* table_padded;
    ngp_half* shard16;             /* (n_chunks x piece) reduce-scatter output (mode 1) */
    float* small;                  /* (n_density + n_rgb) f32 */
    int32_t* flags;                /* 16 x i32, zeroed once by the caller */
    int32_t* step_state;           /* 4 x i32 (ngp_adam_step_field), zeroed / set to the steps taken once by the caller */
    ngp_half* stage;               /* mode 2: (world x piece) f16 landing area of the peers' slices; NULL otherwise */
} ngp_exchange_config;
/* comm NULL (config ignored): back to the single-process tail.  The communicator must outlive the stepper or be detached first. */
int ngp_stepper_set_exchange(ngp_stepper* s, ngp_comm* comm, const ngp_exchange_config* config);
/* lr, step (1-based call count), grad_scale = the total factor the REDUCED gradients carry (loss_scale x world x grad_scale). */
int ngp_stepper_tail(ngp_stepper* s, float lr, int32_t step, float grad_scale, ngp_stream_t main_stream);
/* With stage timing on (ngp_stepper_timing): device time of the last tail's grid exchange, first grid collective -> table gathered
 * (exchange_ms), and the part of it that ran after the table backward had finished, i.e. that nothing on the main stream hid
 * (exposed_ms).  Synchronises.  Negative = not recorded. */
int ngp_stepper_exchange_times(ngp_stepper* s, float* exchange_ms, float* exposed_ms);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
