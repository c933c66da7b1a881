/* Library-internal entry points of libngp_hip.so -- NOT part of the drop-in boundary (that is include/ngp_hip.h).
 *
 * What is here: (1) launchers one translation unit of the library calls in another (the native stepper's and the frame loop's building
 * blocks: composite forward / backward with the loss and the live-sample bookkeeping folded in, list / device-sized forms of the
 * field forward, the table backward that leaves the coarse levels' partial tables to the fused Adam), and (2) diagnostics and
 * white-box hooks the -m gpu tests reach through ctypes (workgroup map, workspace layout, row-major feature transposes, the
 * tiny-cuda-nn-style atomic table backward kept as a cross-check).  They are exported from the shared object so that the tests can
 * call them; a maintainer binding the reference to this library needs none of them.  Same conventions as the public header.
 */
#ifndef NGP_HIP_INTERNAL_H
#define NGP_HIP_INTERNAL_H
#include "../../include/ngp_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The same backward (one launch group) WITHOUT its last launch: the coarse dense levels, whose lists are split over K tasks, are
 * left as K partial f32 tables each in the workspace, described by *partials_out; grad_table holds the gradient of every OTHER
 * level.  ngp_adam_step_field_merge reads the partials itself (same sums in the same order, the same f16 rounding: the update is
 * bit-identical to merge + ngp_adam_step_field).  The record is valid until the workspace is written again. */
typedef struct ngp_grid_partials {
    int32_t n_levels, reserved;                 /* levels 0 .. n_levels-1 (the table's first entries) come as partials */
    int64_t value_end;                          /* 2 x offset[n_levels]: gradient VALUES below this index are not in grad_table */
    uint32_t offset[NGP_MAX_LEVELS + 1];        /* entry offsets of those levels */
    int32_t k_split[NGP_MAX_LEVELS];            /* partial tables per level */
    int64_t part_off[NGP_MAX_LEVELS];           /* entry offset of level l's first partial table inside `partial` (then + k x size) */
    const float* partial;                       /* (entries, 2) f32, device */
} ngp_grid_partials;

/* ngp_field_bwd with the native stepper's overflow guard (what torch.cuda.amp.GradScaler does for the reference under Lightning's
 * precision=16, train.py:274: a step whose gradients hold an inf / NaN is skipped).  nonfinite2 = two i32 flags on the device (NULL:
 * no guard); this call ORs 1 into nonfinite2[parity & 1] when a weight-gradient sum of either network is not finite -- an f16
 * overflow anywhere in the forward recompute or the backward chain ends there -- and clears nonfinite2[(parity & 1) ^ 1], the flag of
 * the next step: the optimizer launch of THIS step reads nonfinite2[parity & 1] as its found_inf.  A feature gradient that leaves the
 * f16 range raises the flag as well (finite in the f32 accumulator, inf in the half the table backward reads).  loss_scale_dev (may be
 * NULL): a factor on loss_scale read from device memory by the launch -- the stepper's dynamic loss scale.  din_limit: the
 * magnitude beyond which a feature gradient raises the flag (0: 65504; 65504 / world under a data-parallel exchange). */
int ngp_field_bwd_guarded(const ngp_half* feats, const float* dirs, const ngp_half* h, const ngp_half* density_w,
                          const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale,
                          const float* loss_scale_dev, float din_limit, int n_samples, const int32_t* active_idx, const int32_t* n_active,
                          ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial, int32_t* nonfinite2, int parity,
                          ngp_stream_t stream);
/* Dynamic loss scale of the native step (round 6; GradScaler's rule on the device, see ngp_stepper_set_loss_scaler): hands the NEXT
 * optimizer launch this thread enqueues through an ngp_adam_step_field* entry point the scaler's device state = {f32 scale[2], i32
 * growth_tracker[2]} and the half (slot 0 / 1) this step's launches read; the launch divides the gradients by scale[slot] on top
 * of grad_scale and writes the next step's scale and tracker into slot ^ 1.  state == NULL: withdraws a pending hand-over. */
int ngp_adam_use_loss_scaler(float* state, int slot, float growth_factor, float backoff_factor, int growth_interval,
                             float min_scale, float max_scale);
/* Before an optimizer launch the caller enqueues itself on gradients a stepper's backward left behind (FusedAdam behind render()):
 * hands the stepper's dynamic loss scale to that launch (ngp_adam_use_loss_scaler) and returns this step's overflow flag in
 * *found_inf (device i32; NULL when no guarded field backward ran since the last update). */
struct ngp_stepper;
int ngp_stepper_before_update(struct ngp_stepper* s, int32_t** found_inf);
/* 1 when ngp_field_bwd reads h / writes dh_scratch (the two-launch A/B build), 0 when both may be NULL (the one-launch kernel, round 6:
 * h is recomputed from the features and dL/dh handed from the colour net to the density net in registers). */
int ngp_field_bwd_uses_h(void);
/* test hook: ngp_field_bwd as the two launches (colour net -> dh_scratch -> density net) whatever the build's default. */
int ngp_field_bwd_two_launches(const ngp_half* feats, const float* dirs, const ngp_half* h, const ngp_half* density_w,
                               const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale,
                               int n_samples, const int32_t* active_idx, const int32_t* n_active,
                               ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial, ngp_stream_t stream);

/* The same launch also draws the marcher's per-ray jitter (custom_functions.py:83: torch.rand_like(rays_o[:, 0])):
 * noise (R) f32 in [0,1) from a counter-based generator keyed by (seed, ray). */
int ngp_ray_aabb_near_noise(const float* rays_o, const float* rays_d,
                            const float* center, const float* half_size, float near_distance,
                            int n_rays, uint64_t seed, float* hits_t, float* noise, ngp_stream_t stream);

/* (_h: the same with the live-sample count also stored to n_active_host, pinned device-mapped host memory, may be NULL) */
int ngp_composite_train_fw_loss_h(const float* sigmas, const float* rgbs, const float* deltas,
                                  const float* ts, const int64_t* rays_a, float T_threshold,
                                  int n_rays, int n_samples, int64_t* total_samples, float* opacity,
                                  float* depth, float* rgb, float* ws, int32_t* ray_offsets,
                                  int32_t* n_active, int32_t* n_active_host, const float* gt_rgb, const float* bg,
                                  float lambda_opacity, float grad_scale, float* loss, float* sq_err,
                                  float* dL_drgb, float* dL_dopacity, void* workspace,
                                  size_t workspace_bytes, ngp_stream_t stream);

/* The same idea for render()'s training branch (rendering.py:121-163), where the caller forms the loss: the forward also writes
 * the BLENDED colour rgb_out (R,3) = rgb + bg (1 - opacity) (bg 3 floats on the device, NULL = black; rgb_out may be NULL) and
 * leaves the rows' live counts in ray_counts; the backward takes its seeds w.r.t. that blended colour (g_opacity may be NULL),
 * folds the blend's backward in, prefixes the counts and writes n_active -- three launches less than ngp_composite_train_fw +
 * ngp_active_scan + ngp_bg_blend and ngp_bg_blend_bw + ngp_composite_train_bw, the same bits. */
int ngp_composite_train_fw_blend(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                 const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                                 int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                                 int32_t* ray_counts, const float* bg, float* rgb_out, ngp_stream_t stream);

int ngp_composite_train_bw_render(const float* g_opacity, const float* g_depth, const float* g_rgb,
                                  const float* g_ws, const float* sigmas, const float* rgbs, const float* ws,
                                  const float* deltas, const float* ts, const int64_t* rays_a,
                                  const float* opacity, const float* depth, const float* rgb, float T_threshold,
                                  int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs,
                                  const int32_t* ray_counts, int32_t* active_idx, const float* xyzs, float* x_active,
                                  int32_t* n_active, const float* bg, ngp_stream_t stream);

/* ... with the gradient of the table's first levels taken from the partial tables ngp_hashgrid_bwd_binned_deferred left behind. */
int ngp_adam_step_field_merge(float* grid_param, ngp_half* grid_param_h, ngp_half* grid_grad,
                              float* grid_m, float* grid_v, int64_t n_grid,
                              float* density_param, ngp_half* density_param_h,
                              const float* density_partials, float* density_m, float* density_v,
                              int n_density,
                              float* rgb_param, ngp_half* rgb_param_h, const float* rgb_partials,
                              float* rgb_m, float* rgb_v, int n_rgb,
                              int n_partials, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale,
                              const int32_t* found_inf, int32_t* step_state, const ngp_grid_partials* partials,
                              ngp_stream_t stream);

int ngp_hashgrid_bwd_binned_deferred(const float* x, const float* xyz_min, const float* xyz_max,
                                     const ngp_half* dfeats, const ngp_grid_meta* meta, int n_samples,
                                     const int32_t* active_idx, const int32_t* n_active,
                                     void* workspace, size_t workspace_bytes, ngp_half* grad_table,
                                     ngp_grid_partials* partials_out, ngp_stream_t stream);

/* same with a device-side sample count (see ngp_hashgrid_fwd_n); n_dev may be NULL */
int ngp_field_fwd_n(const ngp_half* feats, const float* dirs,
                    const ngp_half* density_w, const ngp_half* rgb_w, int n_samples_max,
                    const int32_t* n_dev, float* sigmas, float* rgbs, ngp_half* h_out,
                    ngp_stream_t stream);

/* The native stepper's march as TWO launches: (1) render()'s prologue (one box, near clamp; the jitter draw keyed by (seed, ray)) +
 * pass 1 of the march by the same waves (hits_t (R,2), noise (R), rays_a[:,0] and [:,2], counts (R) i32 = the rays' sample counts,
 * t_scratch); (2) pass 2 whose workgroups prefix `counts` themselves (rays_a[:,1]), write {S, R} to `counter` -- pinned host
 * memory, system-scope stores from the last workgroup BEFORE it expands, so a polling host can size the forward's launches while
 * the expansion runs -- and expand the samples into xyzs / dirs / deltas / ts (caller-allocated for the worst case it accepts;
 * S > capacity is the caller's to check before reading).  The same arithmetic and packing as ngp_ray_aabb_near_noise +
 * ngp_raymarching_train_count + ngp_raymarching_train_write; counts must be 16-byte aligned. */
int ngp_march_train_fused(const float* rays_o, const float* rays_d, const float* center, const float* half_size,
                          float near_distance, uint64_t seed, const uint8_t* density_bitfield, int cascades, float scale,
                          float exp_step_factor, int grid_size, int max_samples, int n_rays,
                          float* hits_t, float* noise, int64_t* rays_a, int32_t* counts, int32_t* counter, float* t_scratch,
                          float* xyzs, float* dirs, float* deltas, float* ts, ngp_stream_t stream);

/* ngp_raymarching_train_count that also prepares the compact first-round list of the two-round forward: offs_k (n_rays, i32) =
 * exclusive scan of min(N, first_k) in ray order, counter[3] = its total (counter then holds 4 x i32).  offs_k NULL: exactly
 * ngp_raymarching_train_count. */
int ngp_raymarching_train_count_k(const float* rays_o, const float* rays_d, const float* hits_t,
                                  const uint8_t* density_bitfield, int cascades, float scale,
                                  float exp_step_factor, const float* noise, int grid_size,
                                  int max_samples, int n_rays, int64_t* rays_a, int32_t* counter,
                                  float* t_scratch, int first_k, int32_t* offs_k, ngp_stream_t stream);

/* ngp_raymarching_train_write that also lists the ids of every ray's first min(N, first_k) samples (first_k in 1..64):
 * list_k[ray * first_k + k] = start + k, or -1 where the ray has fewer (a padded list of n_rays * first_k entries), and clears
 * *n_clear (may be NULL) on the side: the first round of the two-round forward (ngp_stepper, "two-round forward" below). */
int ngp_raymarching_train_write_k(const float* rays_o, const float* rays_d, const int64_t* rays_a,
                                  const float* t_scratch, float scale, float exp_step_factor,
                                  int grid_size, int max_samples, int n_rays,
                                  float* xyzs, float* dirs, float* deltas, float* ts,
                                  int first_k, int32_t* list_k, int32_t* n_clear, ngp_stream_t stream);

/* The same with a COMPACT list in ray order: list_k[offs_k[ray] + k] = start + k for k < min(N, first_k), where offs_k is the
 * exclusive scan of min(N, first_k) over the rays that ngp_raymarching_train_count_k wrote (its total is counter[3]: the list's
 * length).  No padding entries: late in training most rays have no samples at all. */
int ngp_raymarching_train_write_kc(const float* rays_o, const float* rays_d, const int64_t* rays_a,
                                   const float* t_scratch, float scale, float exp_step_factor,
                                   int grid_size, int max_samples, int n_rays,
                                   float* xyzs, float* dirs, float* deltas, float* ts,
                                   int first_k, const int32_t* offs_k, int32_t* list_k, int32_t* n_clear, ngp_stream_t stream);

/* Two-round forward: which rays are still transparent behind their first first_k (<= 64) samples?  For every ray with N > first_k
 * whose transmittance after its first first_k samples is above T_threshold (the composite's own arithmetic on sigmas / deltas at
 * those samples), the ids of its remaining samples are appended to list_rest (*n_rest += N - first_k; order unspecified).  The
 * samples NOT listed lie behind their ray's early stop: volumerendering.cu:20-44 never reads them, so neither does the composite
 * that follows -- the field need not be evaluated there. */
int ngp_composite_probe(const float* sigmas, const float* deltas, const int64_t* rays_a, int first_k, float T_threshold,
                        int n_rays, int32_t* list_rest, int32_t* n_rest, ngp_stream_t stream);

/* The same pair WITHOUT the one-workgroup scan kernel between them (the native step's default): the forward leaves the per-row
 * COUNTS of live samples in ray_counts (R) i32 (16-byte aligned) and the per-row loss terms in the workspace; the backward's
 * workgroups prefix the counts themselves (integer sums: exactly the offsets ngp_composite_train_fw_loss would have written), its
 * last workgroup writes n_active (+ n_active_host, pinned host memory, may be NULL), its first adds the loss terms in a fixed
 * order into loss / sq_err.  n_samples must be > 0 (a batch without samples has no backward: use ngp_composite_train_fw_loss). */
int ngp_composite_train_fw_loss_counts(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                       const int64_t* rays_a, float T_threshold, int n_rays, int n_samples,
                                       int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                                       int32_t* ray_counts, const float* gt_rgb, const float* bg,
                                       float lambda_opacity, float grad_scale, float* dL_drgb,
                                       float* dL_dopacity, void* workspace, size_t workspace_bytes, ngp_stream_t stream);

int ngp_composite_train_bw_tail(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                                const float* dL_dws, const float* sigmas, const float* rgbs, const float* ws,
                                const float* deltas, const float* ts, const int64_t* rays_a,
                                const float* opacity, const float* depth, const float* rgb, float T_threshold,
                                int n_rays, int n_samples, float* dL_dsigmas, float* dL_drgbs,
                                const int32_t* ray_counts, int32_t* active_idx, const float* xyzs, float* x_active,
                                int32_t* n_active, int32_t* n_active_host, float* loss, float* sq_err,
                                const void* workspace, size_t workspace_bytes, ngp_stream_t stream);

/* Encode forward over an explicit list of sample ids: work item j < n_list (n_list_max on the host; min(*n_list_dev, n_list_max) if
 * n_list_dev is given) encodes sample list[j] and writes feats[level][list[j]] (level stride n_samples); entries outside
 * [0, n_samples) are padding and skipped.  Per-sample results, independent of the list order. */
int ngp_hashgrid_fwd_list(const float* x, const float* xyz_min, const float* xyz_max,
                          const ngp_half* table, const ngp_grid_meta* meta, int n_samples,
                          const int32_t* list, int n_list_max, const int32_t* n_list_dev,
                          ngp_half* feats, ngp_stream_t stream);

/* ngp_field_fwd over an explicit list of sample ids (see ngp_hashgrid_fwd_list): inputs are read and sigmas / rgbs / h_out written
 * at the listed samples' own places; feats has level stride n_samples. */
int ngp_field_fwd_list(const ngp_half* feats, const float* dirs,
                       const ngp_half* density_w, const ngp_half* rgb_w, int n_samples,
                       const int32_t* list, int n_list_max, const int32_t* n_list_dev,
                       float* sigmas, float* rgbs, ngp_half* h_out, ngp_stream_t stream);

/* density-only forward whose sigma of sample s is stored at sigmas_out[scatter_idx[s]]
 * (`density_grid_tmp[c, indices] = self.density(xyzs_w)`, networks.py:256-258).  Duplicate indices: the LARGEST value
 * survives (an integer max on the non-negative floats' bit patterns: deterministic; index_put keeps one of them, unspecified
 * on CUDA).  sigmas_out must be zero-filled by the caller. */
int ngp_density_fwd_scatter(const ngp_half* feats, const ngp_half* density_w, int n_samples,
                            const int32_t* scatter_idx, float* sigmas_out, ngp_stream_t stream);

/* Diagnostics (host only, no launch): the workgroup map ngp_hashgrid_fwd uses for n_chunks = ceil(n_samples / 256) chunks per
 * level -- every level with a table above 1 MiB whole on one XCD, the small ones in sixteenths (csrc/hashgrid.hip, FwdMap).
 * Writes up to max_blocks triples (xcd, level, chunk), one per workgroup that has work, and returns their number (16 * n_chunks
 * when every (level, chunk) is covered once); 0 when the table uses the pair map. */
int ngp_debug_hashgrid_fwd_map(const ngp_grid_meta* meta, int n_chunks, int32_t* xcd_level_chunk, int max_blocks);

/* Diagnostics: ngp_render_test_frame's marcher crosses 8^3-cell blocks without an occupied cell in one hop (csrc/march.hip, march_probe:
 * same samples as the cell-by-cell walk).  enabled == 0 makes the calling process walk cell by cell again, != 0 restores the default;
 * tests render the same frames both ways and compare bits, tools time both. */
int ngp_debug_render_block_hops(int enabled);

/* Diagnostics: iterations of ngp_render_test_frame (reference chunking) that have at most max_rays rays alive march one WAVE per ray
 * (render_march_wave_kernel) instead of one thread per ray.  0 = never, a negative value restores the default.  Same samples either way:
 * tests render both ways and compare bits, tools time the crossover. */
int ngp_debug_render_wave_rays(int max_rays);

/* Did the last front() evaluate the field in two rounds? (1 / 0) */
int ngp_stepper_two_rounds(const ngp_stepper* s);

/* Where an update leaves what it evaluated, as byte offsets into the workspace (diagnostics / tests: which cells were drawn, at
 * which jittered positions, with which density): tmp (C, G^3) f32 = sigma scattered by cell index; cell_idx (n) i32 and
 * xyzs (n,3) f32 of the LAST cascade in evaluation order (n = G^3 in warm-up, G^3 / 2 otherwise). */
int ngp_occupancy_update_workspace_layout(int cascades, int grid_size, size_t* tmp_off, size_t* cell_idx_off, size_t* xyzs_off);

/* Layout converters between the level-major feature layout and tcnn's (S,32) row-major one. */
int ngp_feats_to_rowmajor(const ngp_half* feats, int n_levels, int n_samples, ngp_half* out,
                          ngp_stream_t stream);

int ngp_feats_from_rowmajor(const ngp_half* in, int n_levels, int n_samples, ngp_half* feats,
                            ngp_stream_t stream);

/* The same update for a parameter block whose gradient is still spread over n_partials rows of
 * per-workgroup partial sums (n_partials, n) f32 (ngp_*_bwd's wgrad_partial): reduces the column
 * and applies Adam in one launch. */
int ngp_adam_step_partials(float* param, ngp_half* param_h, const float* partials, int n_partials,
                           float* m, float* v, int n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int step, float grad_scale, const int32_t* found_inf,
                           ngp_stream_t stream);

/* The samples that can carry gradient after compositing: the first min(N, total_samples+1) of
 * every ray (later ones have w = 0 exactly, volumerendering.cu:41).  Writes their ids in ray
 * order to active_idx (capacity S) and the count to n_active (device i32); ray_offsets (R) i32
 * receives each ray's offset into the list.  No host sync. */
int ngp_active_samples(const int64_t* rays_a, const int64_t* total_samples, int n_rays,
                       int32_t* ray_offsets, int32_t* active_idx, int32_t* n_active,
                       ngp_stream_t stream);

/* Encode backward w.r.t. the table: scatter-add of w*dL/dfeat into grad_table (total,2) f16
 * (packed f16 atomics, as tiny-cuda-nn) or f32 when grad_is_f32.  Accumulates (caller zeroes). */
int ngp_hashgrid_bwd(const float* x, const float* xyz_min, const float* xyz_max,
                     const ngp_half* dfeats /* [L][S] half2 */, const ngp_grid_meta* meta,
                     int n_samples, void* grad_table, int grad_is_f32, ngp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
