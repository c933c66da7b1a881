// Native driver of one optimisation step (reference: NeRFSystem.training_step, /root/reference/train.py:159-185).
//
// Why this exists: the step is 12-14 kernel launches of 8-100 us each.  Enqueued from Python through ctypes they cost
// ~0.3 ms of host time per step against ~0.37 ms of device time (tools/profile_host.py, round 2), so every further kernel
// gain would have been eaten by the host.  Here the whole sequence is enqueued by three C calls (~50 us of host time), on
// the caller's two HIP streams (main + marching), with reusable events and the one host wait of a step -- the packed
// sample count of the batch's march, written by the scan kernel into pinned host memory -- polled with a deadline.
// No device memory is allocated here; all buffers are the caller's (ngp_step_buffers).
#include "ngp_common.h"
#include "comm.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr int N_MARKS = 9;       // start + 8 main-stream stages

double spin_limit_s() {
    static const double v = [] { const char* e = getenv("NGP_SPIN_TIMEOUT_S"); const double x = e ? atof(e) : 30.0; return x > 0 ? x : 30.0; }();
    return v;
}

}  // namespace

struct ngp_stepper {
    ngp_stepper_config c;
    ngp_step_buffers b;
    hipEvent_t ready[2] = {}, done[2] = {};
    hipEvent_t mark[N_MARKS] = {}, march_t[2][2] = {};       // stage marks (timing); march_t[set][begin/end]
    bool mark_set[N_MARKS] = {}, march_t_set[2] = {};
    int timing = 0;
    int next_set = 0;
    // the marched-but-not-consumed batch
    bool has_pending = false;
    const float* pend_o = nullptr; const float* pend_d = nullptr;
    int pend_set = 0;
    uint64_t marches = 0;
    // state of the step between front() and update()
    int32_t S = 0, n_part = 0, last_set = 0;
    bool binned = false;
    int device = 0;
    // host-side accounting: seconds spent waiting for a march's count (device-bound) and in everything else the entry points do
    double t_wait = 0.0, t_enqueue = 0.0;
    long long n_fronts = 0;
    // the next batch handed to front(): marched behind the stage march_at names (NGP_MARCH_AT, default mlp_fwd)
    const float* next_o = nullptr; const float* next_d = nullptr;
    hipStream_t next_main = nullptr, next_side = nullptr;
    int march_at = 2;
    // two-round forward (include/ngp_hip.h): mode 0 off / 1 on / 2 auto, first K, the auto switch's state, steps run in two rounds
    int two_round_mode = 2, two_round_k = 32;
    bool two_round_active = false, two_rounds = false;
    int set_k[2] = {0, 0};                 // first K the scan of each march record set prepared a compact list for (0: none)
    long long two_round_steps = 0;
    int32_t prev_S = 0;
    // data-parallel exchange (ngp_stepper_set_exchange): the communicator, the plan, and the events the main stream records for
    // the communicator's stream (one per hand-over point of a tail; reused every step)
    ngp_comm* comm = nullptr;
    ngp_exchange_config x = {};
    hipEvent_t ev_small = nullptr, ev_chunk[8] = {}, ev_done = nullptr;
    hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr, ev_m = nullptr;      // timing: first grid collective / table gathered (comm stream), backward done (main)
    bool x_times_set = false;
    long long tails = 0;
    int64_t group_end[16] = {};           // values (2 x entries) the first g + 1 launch groups of the binned backward complete
    // packed samples in two sets (ngp_stepper_set_sample_sets): set k belongs to march record set k, so that the expansion of a
    // prefetched march (pass 2: march_train_write) can run on the marching stream while the running step still reads its own set
    struct SampleSet { float* xyzs; float* dirs; float* deltas; float* ts; };
    SampleSet samples[2] = {};
    bool two_sample_sets = false;
    bool expanded[2] = {false, false};    // record set k: its samples are already written (by the march, on its stream)
    int32_t* counts[2] = {nullptr, nullptr};   // per record set: the rays' sample counts as a dense i32 array (ngp_march_train_fused's prefix input)
    int counts_rays = 0;
    bool same_stream[2] = {false, false};  // record set k was marched on the main stream itself (render() without a next batch)
    // overflow guard of the native step (GradScaler's skip, train.py:274): two device flags used alternately; the field backward of
    // a step raises guard[guard_parity] when a weight-gradient sum is not finite, the optimizer launch of that step reads it
    int32_t* guard = nullptr;
    int guard_parity = 0;
    bool guard_armed = false;              // this step's field backward ran with the guard (consumed by the update)
    bool fused[2] = {false, false};        // record set k was marched by ngp_march_train_fused (count word published by its expansion launch's last workgroup)
    // dynamic loss scale (ngp_stepper_set_loss_scaler): device state {f32 scale[2], i32 growth_tracker[2]}; the step's launches read
    // half scaler_slot, its optimizer launch writes the other half and the host flips scaler_slot behind it
    float* scaler_state = nullptr;
    bool scaler_on = false;
    int scaler_slot = 0;
    float scaler_growth = 2.0f, scaler_backoff = 0.5f, scaler_lo = 9.5367431640625e-07f /* 2^-20 */, scaler_hi = 1.152921504606847e18f /* 2^60: GradScaler has no cap; f32 seeds x 128 x 2^60 stay finite */;
    int scaler_interval = 2000;
};

void destroy_exchange_events(ngp_stepper* s);

namespace {

struct HostTimer {          // accumulates the lifetime of a scope into *acc
    double* acc; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

#define STEP_TRY(expr) do { const int rc_ = (expr); if (rc_ != 0) return rc_; } while (0)
#define STEP_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

int check_buffers(const ngp_stepper_config& c, const ngp_step_buffers& b) {
    if (b.n_rays < 1 || b.cap < b.n_rays || b.max_partials < 1) return NGP_EINVAL;
    const void* need[] = {b.xyzs, b.dirs, b.deltas, b.ts, b.feats, b.h, b.sigmas, b.rgbs, b.ws, b.dL_dsigmas, b.dL_drgbs, b.active, b.dh,
                          b.dfeats, b.total, b.opacity, b.depth, b.rgb, b.dL_drgb, b.dL_dopacity, b.ray_offs, b.zeros, b.n_active, b.stats,
                          b.partials, b.fw_ws, b.hits_t[0], b.hits_t[1], b.rays_a[0], b.rays_a[1], b.noise[0], b.noise[1], b.scratch[0],
                          b.scratch[1], b.counter[0], b.counter[1]};
    for (const void* p : need) if (p == nullptr) return NGP_EINVAL;
    if (b.bin_max > 0 && (b.bin_ws == nullptr || b.x_act == nullptr)) return NGP_EINVAL;
    if (b.distortion && (!b.ws_incl || !b.wts_incl || !b.dL_dws || !b.dist || !b.dist_seed)) return NGP_EINVAL;
    if (c.lambda_distortion > 0 && !b.distortion) return NGP_EINVAL;
    return 0;
}

inline void mark(ngp_stepper* s, int i, hipStream_t st) {
    if (!s->timing) return;
    if (hipEventRecord(s->mark[i], st) == hipSuccess) s->mark_set[i] = true;
}

// Bounded host wait for the march of record set k: first the pinned count word (no HIP call in that loop), then the
// event, which is what orders the caller's following launches behind the march.
int wait_march(ngp_stepper* s, int k, bool count_is_enough = false) {
    volatile int32_t* cnt = s->b.counter[k];
    const auto t_begin = std::chrono::steady_clock::now();
    struct Account { ngp_stepper* s; std::chrono::steady_clock::time_point t0;
                     ~Account() { const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); s->t_wait += d; s->t_enqueue -= d; } } account{s, t_begin};
    const auto t_end = t_begin + std::chrono::duration<double>(spin_limit_s());
    unsigned spins = 0;
    while (cnt[0] < 0) {
        if ((++spins & 1023u) == 0) {
            if (hipEventQuery(s->done[k]) == hipSuccess) break;
            if (std::chrono::steady_clock::now() > t_end) return NGP_ETIMEOUT;
        }
    }
    // marched on the main stream itself by ngp_march_train_fused: what the caller enqueues next is ordered behind the march by the
    // stream, and the count (published before the expansion) is all the host needs -- it sizes the forward's launches while the
    // expansion still runs.  NOT for the scan path: its kernel publishes counter[0] before counter[3] (the launch size of the compact
    // two-round list, which forward_field reads on the host) with plain stores; only the event orders those two reads.
    if (count_is_enough && s->same_stream[k] && s->fused[k] && cnt[0] >= 0) return 0;
    hipError_t e;
    while ((e = hipEventQuery(s->done[k])) == hipErrorNotReady) {
        if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() > t_end) return NGP_ETIMEOUT;
    }
    return e == hipSuccess ? 0 : (int)e;
}

// where in the step the next batch's march is enqueued (it starts behind whatever the main stream has queued by then)
enum { AT_TOP = 0, AT_HASHGRID_FWD, AT_MLP_FWD, AT_COMPOSITE_FW, AT_COMPOSITE_BW, AT_MLP_BWD, AT_HASHGRID_BWD, AT_ADAM };
int march_at_from_env() {
    static const int v = [] {
        const char* e = getenv("NGP_MARCH_AT");
        if (!e) return (int)AT_MLP_FWD;            // profiles/archive_r01_r04/r03_march_sweep.txt: 0.417 ms per step against 0.424 behind the composite forward
        const char* names[] = {"top", "hashgrid_fwd", "mlp_fwd", "composite_fw", "composite_bw", "mlp_bwd", "hashgrid_bwd", "adam"};
        for (int i = 0; i < 8; ++i) if (strcmp(e, names[i]) == 0) return i;
        return (int)AT_MLP_FWD;
    }();
    return v;
}

int ensure_counts(ngp_stepper* s) {
    const int n = s->b.n_rays;
    if (s->counts[0] != nullptr && s->counts_rays >= n) return 0;
    for (int k = 0; k < 2; ++k) { if (s->counts[k]) (void)hipFree(s->counts[k]); s->counts[k] = nullptr; }
    for (int k = 0; k < 2; ++k) STEP_HIP(hipMalloc(reinterpret_cast<void**>(&s->counts[k]), ((size_t)n + 8) * sizeof(int32_t)));
    s->counts_rays = n;
    return 0;
}

int do_march(ngp_stepper* s, const float* rays_o, const float* rays_d, hipStream_t main, hipStream_t side);
inline int march_next_if_at(ngp_stepper* s, int stage) {
    if (s->next_o == nullptr || s->march_at != stage) return 0;
    const float* o = s->next_o; const float* d = s->next_d;
    s->next_o = s->next_d = nullptr;
    return do_march(s, o, d, s->next_main, s->next_side);
}

int do_march(ngp_stepper* s, const float* rays_o, const float* rays_d, hipStream_t main, hipStream_t side) {
    if (s->has_pending) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d);
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    const int k = s->next_set; s->next_set ^= 1;
    if (side != main) {
        STEP_HIP(hipEventRecord(s->ready[k], main));
        STEP_HIP(hipStreamWaitEvent(side, s->ready[k], 0));
    }
    b.counter[k][0] = -1;
    b.counter[k][3] = -1;                  // (a stale two-round launch size of this set's previous march must not pass a range check)
    s->march_t_set[k] = false;
    s->same_stream[k] = side == main;
    s->fused[k] = false;
    if (s->timing) STEP_HIP(hipEventRecord(s->march_t[k][0], side));
    const bool lists_wanted = (s->two_round_mode == 1 || (s->two_round_mode == 2 && s->two_round_active)) && b.list_k && b.list_rest &&
                              b.two_round_counts && b.offs_k[k];
    // (the fused march's expansion sums the counts in front of every 4-ray workgroup itself: R^2 / 4 reads from L2 -- 33 MB at 8192 rays,
    //  2 GB at 65 536, where the scan launch of the other path is cheaper)
    constexpr int FUSED_MARCH_MAX_RAYS = 32768;
    if (s->two_sample_sets && !lists_wanted && b.n_rays <= FUSED_MARCH_MAX_RAYS) {
        // TWO launches: prologue + count, prefix + expansion into this record set's own sample buffers (the running step reads the
        // other set).  The count reaches the pinned word from the second launch's last workgroup before it expands.
        STEP_TRY(ensure_counts(s));
        const ngp_stepper::SampleSet& w = s->samples[k];
        s->set_k[k] = 0;
        STEP_TRY(ngp_march_train_fused(rays_o, rays_d, c.center, c.half_size, c.near_distance, c.noise_seed + 0x9E3779B97F4A7C15ull * (++s->marches),
                                       c.density_bitfield, c.cascades, c.scale, c.exp_step_factor, c.grid_size, c.max_samples, b.n_rays,
                                       b.hits_t[k], b.noise[k], b.rays_a[k], s->counts[k], b.counter[k], b.scratch[k],
                                       w.xyzs, w.dirs, w.deltas, w.ts, (ngp_stream_t)side));
        s->expanded[k] = true;
        s->fused[k] = true;
        if (s->timing) { STEP_HIP(hipEventRecord(s->march_t[k][1], side)); s->march_t_set[k] = true; }
        STEP_HIP(hipEventRecord(s->done[k], side));
        s->has_pending = true; s->pend_o = rays_o; s->pend_d = rays_d; s->pend_set = k;
        return 0;
    }
    STEP_TRY(ngp_ray_aabb_near_noise(rays_o, rays_d, c.center, c.half_size, c.near_distance, b.n_rays, c.noise_seed + 0x9E3779B97F4A7C15ull * (++s->marches),
                                     b.hits_t[k], b.noise[k], (ngp_stream_t)side));
    // (two-round forward: the scan kernel also places every ray's first K samples in the compact first-round list)
    // ... only while the two-round forward is in use: the second scan lengthens the march by ~6 us, which the benchmark's
    // operating point (one round) would pay for nothing
    const bool lists = (s->two_round_mode == 1 || (s->two_round_mode == 2 && s->two_round_active)) && b.list_k && b.list_rest &&
                       b.two_round_counts && b.offs_k[k];
    s->set_k[k] = lists ? s->two_round_k : 0;
    STEP_TRY(ngp_raymarching_train_count_k(rays_o, rays_d, b.hits_t[k], c.density_bitfield, c.cascades, c.scale, c.exp_step_factor, b.noise[k],
                                           c.grid_size, c.max_samples, b.n_rays, b.rays_a[k], b.counter[k], b.scratch[k], s->set_k[k],
                                           lists ? b.offs_k[k] : nullptr, (ngp_stream_t)side));
    // pass 2 on the same stream, into this record set's own sample buffers (the step that is running reads the other set): 9 us of
    // kernel and a launch gap less on the main stream's critical path.  Not for a march whose scan prepared the two-round lists
    // (their expansion kernels also build the first-round list and run where the round is decided: forward_field).
    s->expanded[k] = false;
    if (s->two_sample_sets && side != main && !lists) {
        const ngp_stepper::SampleSet& w = s->samples[k];
        STEP_TRY(ngp_raymarching_train_write(rays_o, rays_d, b.rays_a[k], b.scratch[k], c.scale, c.exp_step_factor, c.grid_size, c.max_samples, b.n_rays,
                                             w.xyzs, w.dirs, w.deltas, w.ts, (ngp_stream_t)side));
        s->expanded[k] = true;
    }
    if (s->timing) { STEP_HIP(hipEventRecord(s->march_t[k][1], side)); s->march_t_set[k] = true; }
    STEP_HIP(hipEventRecord(s->done[k], side));
    s->has_pending = true; s->pend_o = rays_o; s->pend_d = rays_d; s->pend_set = k;
    return 0;
}

}  // namespace

void destroy_exchange_events(ngp_stepper* s) {
    hipEvent_t* all[] = {&s->ev_small, &s->ev_done, &s->ev_x0, &s->ev_x1, &s->ev_m};
    for (hipEvent_t* e : all) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    for (int i = 0; i < 8; ++i) if (s->ev_chunk[i]) { (void)hipEventDestroy(s->ev_chunk[i]); s->ev_chunk[i] = nullptr; }
}

extern "C" {
#pragma GCC visibility push(default)

int ngp_stepper_create(const ngp_stepper_config* config, const ngp_step_buffers* buffers, ngp_stepper** out) {
    if (!config || !buffers || !out) return NGP_EINVAL;
    *out = nullptr;
    const ngp_stepper_config& c = *config;
    // (the f32 masters and Adam moments may be NULL for a stepper that only runs the render-shaped halves: update() then refuses)
    const void* need[] = {c.center, c.half_size, c.xyz_min, c.xyz_max, c.density_bitfield, c.enc_half, c.rgb_half, c.grid_grad16};
    for (const void* p : need) if (p == nullptr) return NGP_EINVAL;
    if (c.cascades < 1 || c.grid_size < 1 || c.max_samples < 1 || c.n_grid < 1 || c.n_density != NGP_DENSITY_NET_PARAMS || c.n_rgb != NGP_RGB_NET_PARAMS)
        return NGP_EINVAL;
    if (c.meta.n_features != 2 || c.meta.n_levels != 16) return NGP_EUNSUP;          // the fused field kernels' configuration
    const int rc = check_buffers(c, *buffers);
    if (rc) return rc;
    ngp_stepper* s = new (std::nothrow) ngp_stepper();
    if (!s) return NGP_EINVAL;
    s->c = c; s->b = *buffers;
    s->march_at = march_at_from_env();
    if (const char* e = getenv("NGP_TWO_ROUND")) s->two_round_mode = strcmp(e, "on") == 0 ? 1 : (strcmp(e, "off") == 0 ? 0 : 2);
    if (const char* e = getenv("NGP_TWO_ROUND_K")) { const int k = atoi(e); if (k >= 1 && k <= 64) s->two_round_k = k; }
    (void)hipGetDevice(&s->device);
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&s->ready[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->done[k], hipEventDisableTiming);
        for (int j = 0; j < 2 && e == hipSuccess; ++j) e = hipEventCreate(&s->march_t[k][j]);
    }
    for (int i = 0; i < N_MARKS && e == hipSuccess; ++i) e = hipEventCreate(&s->mark[i]);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->guard), 2 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(s->guard, 0, 2 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->scaler_state), 4 * sizeof(float));
    if (e != hipSuccess) { ngp_stepper_destroy(s); return (int)e; }
    *out = s;
    return 0;
}

int ngp_stepper_destroy(ngp_stepper* s) {
    if (!s) return 0;
    if (s->has_pending) (void)wait_march(s, s->pend_set);          // bounded: a march that never ends must not hang the teardown
    for (int k = 0; k < 2; ++k) {
        if (s->ready[k]) (void)hipEventDestroy(s->ready[k]);
        if (s->done[k]) (void)hipEventDestroy(s->done[k]);
        for (int j = 0; j < 2; ++j) if (s->march_t[k][j]) (void)hipEventDestroy(s->march_t[k][j]);
    }
    for (int i = 0; i < N_MARKS; ++i) if (s->mark[i]) (void)hipEventDestroy(s->mark[i]);
    for (int k = 0; k < 2; ++k) if (s->counts[k]) (void)hipFree(s->counts[k]);
    if (s->guard) (void)hipFree(s->guard);
    if (s->scaler_state) (void)hipFree(s->scaler_state);
    destroy_exchange_events(s);
    delete s;
    return 0;
}

int ngp_stepper_drop_pending(ngp_stepper* s) {
    if (!s) return NGP_EINVAL;
    if (!s->has_pending) return 0;
    // its kernels may still be running on the marching stream and they write the record's buffers
    const int rc = wait_march(s, s->pend_set);
    s->next_set = s->pend_set;          // hand the set back
    s->has_pending = false;
    return rc;
}

int ngp_stepper_set_buffers(ngp_stepper* s, const ngp_step_buffers* buffers) {
    if (!s || !buffers) return NGP_EINVAL;
    STEP_TRY(ngp_stepper_drop_pending(s));
    STEP_TRY(check_buffers(s->c, *buffers));
    s->b = *buffers;
    s->two_sample_sets = false; s->expanded[0] = s->expanded[1] = false;     // the second sample set belonged to the old buffers
    s->S = 0; s->n_part = 0;
    s->set_k[0] = s->set_k[1] = 0;                       // (no march of the new record sets has prepared a first-round list)
    s->two_round_active = false; s->two_rounds = false;  // the auto switch starts over: its evidence (live fraction of the previous
    s->prev_S = 0;                                       //  step, counter[.][2] of the OLD pinned words) does not describe these buffers
    return 0;
}

int ngp_stepper_set_sample_sets(ngp_stepper* s, float* xyzs1, float* dirs1, float* deltas1, float* ts1) {
    if (!s) return NGP_EINVAL;
    STEP_TRY(ngp_stepper_drop_pending(s));               // a pending march may be writing into the set that goes away
    const bool all = xyzs1 && dirs1 && deltas1 && ts1, none = !xyzs1 && !dirs1 && !deltas1 && !ts1;
    if (!all && !none) return NGP_EINVAL;
    if (s->two_sample_sets) {                            // back to the caller's ngp_step_buffers pointers first
        s->b.xyzs = s->samples[0].xyzs; s->b.dirs = s->samples[0].dirs; s->b.deltas = s->samples[0].deltas; s->b.ts = s->samples[0].ts;
    }
    s->expanded[0] = s->expanded[1] = false;
    s->two_sample_sets = all;
    if (all) {
        NGP_CHECK_PTR(xyzs1); NGP_CHECK_PTR(dirs1); NGP_CHECK_PTR(deltas1); NGP_CHECK_PTR(ts1);
        s->samples[0] = {s->b.xyzs, s->b.dirs, s->b.deltas, s->b.ts};
        s->samples[1] = {xyzs1, dirs1, deltas1, ts1};
    }
    return 0;
}

int ngp_stepper_pending(const ngp_stepper* s, const float* rays_o, const float* rays_d) {
    return (s && s->has_pending && s->pend_o == rays_o && s->pend_d == rays_d) ? 1 : 0;
}

int ngp_stepper_last_set(const ngp_stepper* s) { return s ? s->last_set : 0; }
int ngp_stepper_two_rounds(const ngp_stepper* s) { return (s && s->two_rounds) ? 1 : 0; }
int ngp_stepper_record_bytes(int which) {
    return which == 0 ? (int)sizeof(ngp_stepper_config) : which == 1 ? (int)sizeof(ngp_step_buffers) : which == 2 ? (int)sizeof(ngp_exchange_config) : NGP_EINVAL;
}

int ngp_stepper_march(ngp_stepper* s, const float* rays_o, const float* rays_d, ngp_stream_t main_stream, ngp_stream_t march_stream) {
    if (!s) return NGP_EINVAL;
    HostTimer host_timer(&s->t_enqueue);
    return do_march(s, rays_o, rays_d, ngp_stream(main_stream), ngp_stream(march_stream));
}

// march hand-over -> sample expansion -> hash grid -> field: the part both step shapes share.  Leaves S in s->S.
static int forward_field(ngp_stepper* s, const float* rays_o, const float* rays_d, hipStream_t main, ngp_stream_t main_stream, int* k_out) {
    if (!s->has_pending || s->pend_o != rays_o || s->pend_d != rays_d) return NGP_EINVAL;     // march() this batch first
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    const int k = s->pend_set;
    // the step's only host wait.  No hipStreamWaitEvent on the main stream: the host has observed the event, so everything
    // enqueued from here on is ordered behind the march (the barrier packet measured ~20 us of idle main stream per step).
    // On NGP_ETIMEOUT the march stays pending: its kernels may still be writing record set k, so the set must not be handed to
    // another march (do_march refuses while one is pending; a later front() of the same batch waits again).
    STEP_TRY(wait_march(s, k, true));
    s->has_pending = false;
    const int32_t S = b.counter[k][0];
    if (S < 0 || (int64_t)S > b.cap) return NGP_EINVAL;
    s->S = S; s->last_set = k; s->n_part = 0;
    *k_out = k;
    if (s->two_sample_sets) {              // every later stage of this step reads the samples through s->b
        s->b.xyzs = s->samples[k].xyzs; s->b.dirs = s->samples[k].dirs; s->b.deltas = s->samples[k].deltas; s->b.ts = s->samples[k].ts;
    }
    const int n = b.n_rays;
    for (int i = 0; i < N_MARKS; ++i) s->mark_set[i] = false;
    mark(s, 0, main);
    STEP_TRY(march_next_if_at(s, AT_TOP));
    // two rounds?  (auto: from the live fraction of the previous step, which its composite left in pinned memory)
    bool two = false;
    if (s->two_round_mode != 0 && b.list_k && b.list_rest && b.two_round_counts && c.lambda_distortion <= 0 && S > 0) {
        if (s->two_round_mode == 1) two = true;
        else {
            // (the previous step normally consumed the OTHER record set; after a dropped march it was this one, whose word front()
            //  has not reset yet either -- both hold a live count of an earlier step or -1, and a stale estimate only delays the switch)
            const int32_t prev_live = b.counter[k ^ 1][2];
            if (s->prev_S > 0 && prev_live >= 0) {
                const float frac = (float)prev_live / (float)s->prev_S;
                if (frac < 0.15f) s->two_round_active = true; else if (frac > 0.25f) s->two_round_active = false;
            }
            two = s->two_round_active;
        }
    }
    s->prev_S = S;
    s->two_rounds = two;
    const ngp_half* table = c.enc_half + c.n_density;
    ngp_half* h_store = ngp_field_bwd_uses_h() ? b.h : nullptr;      // the one-launch field backward recomputes h: the forward need not store it
    if (!two) {
        if (!s->expanded[k])
            STEP_TRY(ngp_raymarching_train_write(rays_o, rays_d, b.rays_a[k], b.scratch[k], c.scale, c.exp_step_factor, c.grid_size, c.max_samples, n,
                                                 b.xyzs, b.dirs, b.deltas, b.ts, main_stream));
        mark(s, 1, main);
        if (S > 0) {
            STEP_TRY(ngp_hashgrid_fwd(b.xyzs, c.xyz_min, c.xyz_max, table, &c.meta, S, b.feats, main_stream));
            mark(s, 2, main);
            STEP_TRY(march_next_if_at(s, AT_HASHGRID_FWD));
            STEP_TRY(ngp_field_fwd(b.feats, b.dirs, c.enc_half, c.rgb_half, S, b.sigmas, b.rgbs, h_store, main_stream));
            mark(s, 3, main);
            STEP_TRY(march_next_if_at(s, AT_MLP_FWD));
        }
        return 0;
    }
    // round 1: every ray's first K samples (a compact list in ray order where the march's scan prepared it, else a padded one);
    // round 2: the rest of the rays that are still transparent behind them
    const int K = s->two_round_k;
    ++s->two_round_steps;
    int32_t* n2 = b.two_round_counts;
    const bool compact = s->set_k[k] == K && b.offs_k[k] != nullptr;       // this record set's scan placed the first K of every ray
    int n1 = n * K;                                                        // padded list: -1 where a ray has fewer than K samples
    if (compact) {
        n1 = b.counter[k][3];
        if (n1 < 0 || n1 > n * K || n1 > S) return NGP_EINVAL;
        STEP_TRY(ngp_raymarching_train_write_kc(rays_o, rays_d, b.rays_a[k], b.scratch[k], c.scale, c.exp_step_factor, c.grid_size, c.max_samples, n,
                                                b.xyzs, b.dirs, b.deltas, b.ts, K, b.offs_k[k], b.list_k, n2, main_stream));
    } else {
        STEP_TRY(ngp_raymarching_train_write_k(rays_o, rays_d, b.rays_a[k], b.scratch[k], c.scale, c.exp_step_factor, c.grid_size, c.max_samples, n,
                                               b.xyzs, b.dirs, b.deltas, b.ts, K, b.list_k, n2, main_stream));
    }
    mark(s, 1, main);
    STEP_TRY(ngp_hashgrid_fwd_list(b.xyzs, c.xyz_min, c.xyz_max, table, &c.meta, S, b.list_k, n1, nullptr, b.feats, main_stream));
    mark(s, 2, main);
    STEP_TRY(march_next_if_at(s, AT_HASHGRID_FWD));
    STEP_TRY(ngp_field_fwd_list(b.feats, b.dirs, c.enc_half, c.rgb_half, S, b.list_k, n1, nullptr, b.sigmas, b.rgbs, h_store, main_stream));
    STEP_TRY(ngp_composite_probe(b.sigmas, b.deltas, b.rays_a[k], K, c.T_threshold, n, b.list_rest, n2, main_stream));
    STEP_TRY(ngp_hashgrid_fwd_list(b.xyzs, c.xyz_min, c.xyz_max, table, &c.meta, S, b.list_rest, S, n2, b.feats, main_stream));
    STEP_TRY(ngp_field_fwd_list(b.feats, b.dirs, c.enc_half, c.rgb_half, S, b.list_rest, S, n2, b.sigmas, b.rgbs, h_store, main_stream));
    mark(s, 3, main);
    STEP_TRY(march_next_if_at(s, AT_MLP_FWD));
    return 0;
}

// composite backward (+ distortion) -> field backward on the live samples, from seeds w.r.t. the composited per-ray values
enum { TAIL_NONE = 0, TAIL_LOSS = 1, TAIL_RENDER = 2 };   // who turns the rows' live counts into offsets: ngp_active_scan / the tail kernel (NONE) or the backward itself
static int backward_field(ngp_stepper* s, const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws_in,
                          float loss_scale, hipStream_t main, ngp_stream_t main_stream, int32_t* n_partials, int fused_tail = TAIL_NONE) {
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    const int k = s->last_set, n = b.n_rays;
    const int32_t S = s->S;
    *n_partials = 0;
    if (S <= 0) return 0;
    const float* dL_dws = dL_dws_in;
    if (c.lambda_distortion > 0) {
        // losses.py:6-37,58-59: lambda * distortion per ray, mean over rays; its gradient enters the composite as dL/dws
        if (dL_dws_in != nullptr) return NGP_EINVAL;                      // (one source of dL/dws)
        STEP_TRY(ngp_distortion_loss_fw(b.ws, b.deltas, b.ts, b.rays_a[k], n, S, b.dist, b.ws_incl, b.wts_incl, main_stream));
        STEP_TRY(ngp_distortion_loss_bw(b.dist_seed, b.ws_incl, b.wts_incl, b.ws, b.deltas, b.ts, b.rays_a[k], n, S, b.dL_dws, main_stream));
        dL_dws = b.dL_dws;
    }
    // backward only over the samples up to each ray's early stop (the rest have zero gradient); the binned table backward
    // reads the live samples' positions as a stream: composite_bw copies them in list order
    s->binned = S <= b.bin_max;
    if (fused_tail == TAIL_RENDER) {
        // (render()'s branch: seeds w.r.t. the BLENDED colour; the blend's backward, the prefix of the live counts and n_active in the same launch)
        STEP_TRY(ngp_composite_train_bw_render(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, b.sigmas, b.rgbs, b.ws, b.deltas, b.ts, b.rays_a[k], b.opacity,
                                               b.depth, b.rgb, c.T_threshold, n, S, b.dL_dsigmas, b.dL_drgbs, b.ray_offs, b.active,
                                               s->binned ? b.xyzs : nullptr, s->binned ? b.x_act : nullptr, b.n_active, c.bg, main_stream));
    } else if (fused_tail == TAIL_LOSS) {
        // (ray_offs holds the rows' live-sample COUNTS: the backward's workgroups prefix them, write n_active and add the loss terms)
        STEP_TRY(ngp_composite_train_bw_tail(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, b.sigmas, b.rgbs, b.ws, b.deltas, b.ts, b.rays_a[k], b.opacity,
                                             b.depth, b.rgb, c.T_threshold, n, S, b.dL_dsigmas, b.dL_drgbs, b.ray_offs, b.active,
                                             s->binned ? b.xyzs : nullptr, s->binned ? b.x_act : nullptr, b.n_active, b.counter[k] + 2,
                                             b.stats, b.stats + 1, b.fw_ws, b.fw_bytes, main_stream));
    } else {
        STEP_TRY(ngp_composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, b.sigmas, b.rgbs, b.ws, b.deltas, b.ts, b.rays_a[k], b.opacity,
                                        b.depth, b.rgb, c.T_threshold, n, S, b.dL_dsigmas, b.dL_drgbs, b.ray_offs, b.active,
                                        s->binned ? b.xyzs : nullptr, s->binned ? b.x_act : nullptr, main_stream));
    }
    mark(s, 5, main);
    STEP_TRY(march_next_if_at(s, AT_COMPOSITE_BW));
    const int n_part = ngp_field_bwd_partials(S);
    if (n_part < 1 || n_part > b.max_partials) return NGP_EINVAL;
    s->guard_parity ^= 1;
    STEP_TRY(ngp_field_bwd_guarded(b.feats, b.dirs, b.h, c.enc_half, c.rgb_half, b.dL_dsigmas, b.dL_drgbs, loss_scale,
                                   s->scaler_on ? s->scaler_state + s->scaler_slot : nullptr,
                                   s->comm ? 65504.0f / (float)s->comm->world : 0.f,      // (data parallel: the ranks' f16 gradients are summed)
                                   S, b.active, b.n_active,
                                   b.dh, b.dfeats, b.partials, s->guard, s->guard_parity, main_stream));
    s->guard_armed = true;
    mark(s, 6, main);
    STEP_TRY(march_next_if_at(s, AT_MLP_BWD));
    s->n_part = n_part;
    *n_partials = n_part;
    return 0;
}

int ngp_stepper_front(ngp_stepper* s, const float* rays_o, const float* rays_d, const float* rgb_gt,
                      const float* next_o, const float* next_d, float loss_scale, float grad_scale,
                      ngp_stream_t main_stream, ngp_stream_t march_stream, int32_t* n_samples, int32_t* n_partials) {
    if (!s || !n_samples || !n_partials) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d); NGP_CHECK_PTR(rgb_gt);
    HostTimer host_timer(&s->t_enqueue);
    if ((next_o == nullptr) != (next_d == nullptr)) return NGP_EINVAL;
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    hipStream_t main = ngp_stream(main_stream), side = ngp_stream(march_stream);
    int k = 0;
    *n_samples = 0; *n_partials = 0;
    ++s->n_fronts;
    s->next_o = next_o; s->next_d = next_d; s->next_main = main; s->next_side = side;
    STEP_TRY(forward_field(s, rays_o, rays_d, main, main_stream, &k));
    const int32_t S = s->S;
    *n_samples = S;
    const int n = b.n_rays;
    // composite + per-ray loss seeds, then one small kernel: offsets of the live samples and the loss sums
    b.counter[k][2] = -1;
    const bool fused_tail = S > 0;              // composite forward / backward without a scan kernel between them (a batch without samples: the plain pair)
    if (fused_tail) {
        STEP_TRY(ngp_composite_train_fw_loss_counts(b.sigmas, b.rgbs, b.deltas, b.ts, b.rays_a[k], c.T_threshold, n, S, b.total, b.opacity, b.depth, b.rgb,
                                                    b.ws, b.ray_offs, rgb_gt, c.bg, c.lambda_opacity, grad_scale, b.dL_drgb, b.dL_dopacity, b.fw_ws,
                                                    b.fw_bytes, main_stream));
    } else {
        STEP_TRY(ngp_composite_train_fw_loss_h(b.sigmas, b.rgbs, b.deltas, b.ts, b.rays_a[k], c.T_threshold, n, S, b.total, b.opacity, b.depth, b.rgb,
                                               b.ws, b.ray_offs, b.n_active, b.counter[k] + 2, rgb_gt, c.bg, c.lambda_opacity, grad_scale, b.stats,
                                               b.stats + 1, b.dL_drgb, b.dL_dopacity, b.fw_ws, b.fw_bytes, main_stream));
    }
    mark(s, 4, main);
    STEP_TRY(march_next_if_at(s, AT_COMPOSITE_FW));
    STEP_TRY(backward_field(s, b.dL_dopacity, b.zeros, b.dL_drgb, nullptr, loss_scale, main, main_stream, n_partials, fused_tail ? TAIL_LOSS : TAIL_NONE));
    if (s->next_o != nullptr && (S <= 0 || s->march_at < AT_HASHGRID_BWD)) {      // a batch without samples skips the stages a placement may name
        const float* o = s->next_o; const float* d = s->next_d;
        s->next_o = s->next_d = nullptr;
        STEP_TRY(do_march(s, o, d, main, side));
    }
    return 0;
}

// The same step for a caller that forms the loss itself (render()'s training branch, rendering.py:121-163, followed by
// NeRFLoss and autograd): forward half ...
int ngp_stepper_render_forward(ngp_stepper* s, const float* rays_o, const float* rays_d, const float* next_o, const float* next_d,
                               float* rgb_out, ngp_stream_t main_stream, ngp_stream_t march_stream, int32_t* n_samples) {
    if (!s || !n_samples) return NGP_EINVAL;
    NGP_CHECK_PTR(rays_o); NGP_CHECK_PTR(rays_d);
    HostTimer host_timer(&s->t_enqueue);
    if ((next_o == nullptr) != (next_d == nullptr)) return NGP_EINVAL;
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    hipStream_t main = ngp_stream(main_stream), side = ngp_stream(march_stream);
    int k = 0;
    *n_samples = 0;
    STEP_TRY(forward_field(s, rays_o, rays_d, main, main_stream, &k));
    *n_samples = s->S;
    const int n = b.n_rays;
    b.counter[k][2] = -1;                                    // (no live-sample count from this composite: the auto switch stays where it is)
    // ONE launch: composite + the blended colour (rendering.py:153-161); the rows' live counts stay counts for render_backward
    STEP_TRY(ngp_composite_train_fw_blend(b.sigmas, b.rgbs, b.deltas, b.ts, b.rays_a[k], c.T_threshold, n, s->S, b.total, b.opacity, b.depth, b.rgb,
                                          b.ws, b.ray_offs, c.bg, rgb_out, main_stream));
    mark(s, 4, main);
    if (next_o) STEP_TRY(do_march(s, next_o, next_d, main, side));      // (the render-shaped halves: always behind the composite forward)
    return 0;
}

// ... and backward half: g_rgb (R,3) w.r.t. the BLENDED colour, g_opacity / g_depth (R) and g_ws (S) w.r.t. the composited values
// (any of the last three may be NULL = zero).  Leaves the per-workgroup weight-gradient partials like front(); the table
// backward and the update follow as separate calls.
int ngp_stepper_render_backward(ngp_stepper* s, const float* g_rgb, const float* g_opacity, const float* g_depth, const float* g_ws,
                                float loss_scale, ngp_stream_t main_stream, int32_t* n_partials) {
    if (!s || !n_partials) return NGP_EINVAL;
    NGP_CHECK_PTR(g_rgb);
    HostTimer host_timer(&s->t_enqueue);
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    *n_partials = 0;
    if (s->S <= 0) return 0;
    const int n = b.n_rays;
    // the forward left live COUNTS in ray_offs: the backward folds the background blend, their prefix and n_active in
    return backward_field(s, g_opacity, g_depth ? g_depth : b.zeros, g_rgb, g_ws, loss_scale, ngp_stream(main_stream), main_stream, n_partials, TAIL_RENDER);
}

// One launch group of the binned table backward on the main stream.
static int table_backward_group(ngp_stepper* s, int n_groups, int group, ngp_grid_partials* partials_out, ngp_stream_t main_stream) {
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    if (partials_out) return ngp_hashgrid_bwd_binned_deferred(b.x_act, c.xyz_min, c.xyz_max, b.dfeats, &c.meta, s->S, nullptr, b.n_active, b.bin_ws,
                                                              b.bin_bytes, c.grid_grad16, partials_out, main_stream);
    return ngp_hashgrid_bwd_binned_group(b.x_act, c.xyz_min, c.xyz_max, b.dfeats, &c.meta, s->S, nullptr, b.n_active, b.bin_ws, b.bin_bytes,
                                         c.grid_grad16, n_groups, group, main_stream);
}

int ngp_stepper_table_backward(ngp_stepper* s, int n_groups, int group, ngp_stream_t main_stream) {
    if (!s || n_groups < 1 || group < 0 || group >= n_groups) return NGP_EINVAL;
    HostTimer host_timer(&s->t_enqueue);
    if (s->S <= 0) return 0;
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    if (s->binned) {
        STEP_TRY(table_backward_group(s, n_groups, group, nullptr, main_stream));
    } else if (group == 0) {
        STEP_TRY(ngp_hashgrid_bwd_sliced(b.xyzs, c.xyz_min, c.xyz_max, b.dfeats, &c.meta, s->S, b.active, b.n_active, c.grid_grad16, main_stream));
    }
    if (group == n_groups - 1) {
        mark(s, 7, ngp_stream(main_stream));
        STEP_TRY(march_next_if_at(s, AT_HASHGRID_BWD));
    }
    return 0;
}

// The optimizer launch enqueued next divides by this step's dynamic loss scale and writes the next step's (optim.hip: LossScaler).
static void withdraw_scaler() { (void)ngp_adam_use_loss_scaler(nullptr, 0, 2.0f, 0.5f, 1, 1.0f, 1.0f); }

static int hand_over_scaler(ngp_stepper* s) {
    if (!s->scaler_on) return 0;
    STEP_TRY(ngp_adam_use_loss_scaler(s->scaler_state, s->scaler_slot, s->scaler_growth, s->scaler_backoff, s->scaler_interval, s->scaler_lo, s->scaler_hi));
    s->scaler_slot ^= 1;                 // the next step's field backward reads what this launch writes
    return 0;
}

// (csrc/ngp_internal.h) For an optimizer launch the CALLER enqueues (ngp_pl_amd.optim.FusedAdam behind render()'s native backward:
// ngp_adam_step_field on the gradients this stepper left in its buffers): hands that launch the dynamic loss scale, and returns
// the overflow flag this step's field backward may have raised (NULL: no guarded backward since the last update).
int ngp_stepper_before_update(ngp_stepper* s, int32_t** found_inf) {
    if (!s || !found_inf) return NGP_EINVAL;
    *found_inf = s->guard_armed ? s->guard + s->guard_parity : nullptr;
    s->guard_armed = false;
    return hand_over_scaler(s);
}

int ngp_stepper_update(ngp_stepper* s, float lr, int32_t step, float grad_scale, const float* density_partials,
                       const float* rgb_partials, int32_t n_partials, const int32_t* found_inf, int32_t* step_state, ngp_stream_t main_stream) {
    if (!s || step < 1 || (density_partials == nullptr) != (rgb_partials == nullptr)) return NGP_EINVAL;
    HostTimer host_timer(&s->t_enqueue);
    const ngp_stepper_config& c = s->c;
    if (!c.enc_param || !c.enc_m || !c.enc_v || !c.rgb_param || !c.rgb_m || !c.rgb_v) return NGP_EINVAL;      // built without an optimizer state
    const ngp_step_buffers& b = s->b;
    if (density_partials == nullptr) {
        if (s->n_part < 1) return NGP_EINVAL;            // no backward ran (S == 0): nothing to apply
        density_partials = b.partials;
        rgb_partials = b.partials + (size_t)s->n_part * c.n_density;
        n_partials = s->n_part;
    }
    if (n_partials < 1) return NGP_EINVAL;
    if (found_inf == nullptr && s->guard_armed && density_partials == b.partials) found_inf = s->guard + s->guard_parity;    // this step's own field backward
    s->guard_armed = false;
    STEP_TRY(hand_over_scaler(s));
    {
        const int rc_adam = ngp_adam_step_field(c.enc_param + c.n_density, c.enc_half + c.n_density, c.grid_grad16, c.enc_m + c.n_density, c.enc_v + c.n_density, c.n_grid,
                                 c.enc_param, c.enc_half, density_partials, c.enc_m, c.enc_v, c.n_density,
                                 c.rgb_param, c.rgb_half, rgb_partials, c.rgb_m, c.rgb_v, c.n_rgb,
                                 n_partials, lr, c.beta1, c.beta2, c.eps, c.weight_decay, step, grad_scale, 0, found_inf, step_state, main_stream);
        withdraw_scaler();                       // (consumed by a launch that went out; withdrawn if validation refused it)
        if (rc_adam) return rc_adam;
    }
    mark(s, 8, ngp_stream(main_stream));
    STEP_TRY(march_next_if_at(s, AT_ADAM));
    return 0;
}

int ngp_stepper_backward_update(ngp_stepper* s, float lr, int32_t step, float grad_scale, int32_t* step_state, ngp_stream_t main_stream) {
    if (!s || step < 1) return NGP_EINVAL;
    if (s->comm) return NGP_EINVAL;                              // data parallel: ngp_stepper_tail
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    if (s->S <= 0 || !s->binned || s->n_part < 1) {
        STEP_TRY(ngp_stepper_table_backward(s, 1, 0, main_stream));
        return s->S > 0 ? ngp_stepper_update(s, lr, step, grad_scale, nullptr, nullptr, 0, nullptr, step_state, main_stream) : 0;
    }
    HostTimer host_timer(&s->t_enqueue);
    if (!c.enc_param || !c.enc_m || !c.enc_v || !c.rgb_param || !c.rgb_m || !c.rgb_v) return NGP_EINVAL;
    ngp_grid_partials gp;
    const int64_t n_streamed = c.n_grid;                         // the whole table's gradient is applied by the streaming launch
    {
        const int rc = table_backward_group(s, 1, 0, &gp, main_stream);
        if (rc) return rc;
    }
    mark(s, 7, ngp_stream(main_stream));
    STEP_TRY(march_next_if_at(s, AT_HASHGRID_BWD));
    STEP_TRY(hand_over_scaler(s));
    {
        const int rc_adam = ngp_adam_step_field_merge(c.enc_param + c.n_density, c.enc_half + c.n_density, c.grid_grad16, c.enc_m + c.n_density, c.enc_v + c.n_density, n_streamed,
                                       c.enc_param, c.enc_half, b.partials, c.enc_m, c.enc_v, c.n_density,
                                       c.rgb_param, c.rgb_half, b.partials + (size_t)s->n_part * c.n_density, c.rgb_m, c.rgb_v, c.n_rgb,
                                       s->n_part, lr, c.beta1, c.beta2, c.eps, c.weight_decay, step, grad_scale,
                                       s->guard_armed ? s->guard + s->guard_parity : nullptr, step_state, &gp, main_stream);
        withdraw_scaler();                       // (consumed by a launch that went out; withdrawn if validation refused it)
        if (rc_adam) return rc_adam;
    }
    s->guard_armed = false;
    mark(s, 8, ngp_stream(main_stream));
    STEP_TRY(march_next_if_at(s, AT_ADAM));
    return 0;
}

int ngp_stepper_set_exchange(ngp_stepper* s, ngp_comm* comm, const ngp_exchange_config* config) {
    if (!s) return NGP_EINVAL;
    if (comm == nullptr) {
        if (s->comm) (void)hipStreamSynchronize(s->comm->stream);
        s->comm = nullptr;
        destroy_exchange_events(s);
        return 0;
    }
    if (!config) return NGP_EINVAL;
    const ngp_exchange_config& x = *config;
    const ngp_stepper_config& c = s->c;
    if (x.mode < 0 || x.mode > 2 || (x.mode == 2 && (x.n_chunks != 1 || !x.stage)) || x.n_chunks < 1 || x.n_chunks > 8 || x.n_groups < 1 || x.n_groups > 16 || x.piece < 8 || (x.piece & 7)) return NGP_EINVAL;
    if ((int64_t)x.n_chunks * comm->world * x.piece < c.n_grid || (c.n_grid & 15)) return NGP_EINVAL;
    if (x.grad_padded != c.grid_grad16 || x.table_padded != c.enc_half + c.n_density) return NGP_EINVAL;      // the padded storages must be the stepper's own
    if (!x.small || !x.flags || !x.step_state || (x.mode >= 1 && !x.shard16)) return NGP_EINVAL;
    if (!c.enc_param || !c.enc_m || !c.enc_v || !c.rgb_param || !c.rgb_m || !c.rgb_v) return NGP_EINVAL;
    // which chunk may leave behind which launch group: the groups complete contiguous entry ranges in table order
    for (int g = 0; g < x.n_groups; ++g) {
        int64_t a = 0, b = 0;
        STEP_TRY(ngp_hashgrid_bwd_binned_group_entries(&c.meta, 1, x.n_groups, g, &a, &b));
        s->group_end[g] = 2 * b;
    }
    if (s->group_end[x.n_groups - 1] != c.n_grid) return NGP_EINVAL;
    destroy_exchange_events(s);
    hipError_t e = hipEventCreateWithFlags(&s->ev_small, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming);
    for (int i = 0; i < x.n_chunks && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s->ev_chunk[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&s->ev_x0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev_x1);
    if (e == hipSuccess) e = hipEventCreate(&s->ev_m);
    if (e != hipSuccess) { destroy_exchange_events(s); return (int)e; }
    // the caller's flag words outlive a re-attachment (mode switch, stepper rebuild) while the parity count restarts: a flag raised
    // by the last step before the reset must not be read again as this step's (ngp_found_inf2 only ORs into the current set)
    e = hipMemsetAsync(x.flags, 0, 16 * sizeof(int32_t), comm->stream);
    if (e != hipSuccess) { destroy_exchange_events(s); return (int)e; }
    s->comm = comm; s->x = x; s->tails = 0; s->x_times_set = false;
    return 0;
}

// One chunk of the grid gradient to the communicator's stream: behind `ev` (recorded on the main stream by the caller).
static int exchange_chunk(ngp_stepper* s, int chunk, hipStream_t cs) {
    const ngp_exchange_config& x = s->x;
    const int64_t C = (int64_t)s->comm->world * x.piece;
    ngp_half* g = x.grad_padded + (size_t)chunk * C;
    if (x.mode == 1) return ngp_comm_reduce_scatter(s->comm, g, x.shard16 + (size_t)chunk * x.piece, x.piece, NGP_COMM_F16, (ngp_stream_t)cs);
    if (x.mode == 2) {          // direct: the peers' slices of this rank's share over all links at once, then the sum in rank order (f32)
        STEP_TRY(ngp_comm_exchange_slices(s->comm, g, x.stage, x.piece, NGP_COMM_F16, (ngp_stream_t)cs));
        return ngp_sum_slices_f16(g + (size_t)s->comm->rank * x.piece, x.stage, s->comm->world, s->comm->rank, x.piece, x.shard16, (ngp_stream_t)cs);
    }
    return ngp_comm_all_reduce(s->comm, g, C, NGP_COMM_F16, (ngp_stream_t)cs);
}

// Data parallel + dynamic loss scale: a rank whose own field backward raised the overflow flag makes every rank see it -- one inf into
// the MLP sums it is about to all-reduce.  The reduced sums are what the skip decision and the scale's backoff are keyed on, so all
// ranks skip and halve together (a flag each rank kept to itself would let their scales part).
// The second flag is LAST step's verdict on this rank's share of the reduced table gradient (sharded / direct modes: the f16 sum of the
// ranks' tables can overflow although no rank's own entries did; only the share's owner sees it, one step late for everybody else):
// its owner skipped that block then, everyone skips and backs off now.
__global__ void poison_small_kernel(float* __restrict__ small, const int32_t* __restrict__ flag, const int32_t* __restrict__ last_shard_flag) {
    if ((flag != nullptr && *flag != 0) || (last_shard_flag != nullptr && *last_shard_flag != 0)) small[0] = __builtin_inff();
}

int ngp_stepper_tail(ngp_stepper* s, float lr, int32_t step, float grad_scale, ngp_stream_t main_stream) {
    if (!s || step < 1 || grad_scale == 0.f) return NGP_EINVAL;
    if (!s->comm) return NGP_EINVAL;                                  // no exchange installed: table_backward() + update()
    HostTimer host_timer(&s->t_enqueue);
    const ngp_stepper_config& c = s->c;
    const ngp_step_buffers& b = s->b;
    const ngp_exchange_config& x = s->x;
    ngp_comm* comm = s->comm;
    hipStream_t main = ngp_stream(main_stream), cs = comm->stream;
    const int64_t C = (int64_t)comm->world * x.piece, padded = (int64_t)x.n_chunks * C;
    const int32_t S = s->S;
    const bool timing = s->timing != 0;
    s->x_times_set = false;
    // (1) the MLP blocks: per-workgroup partial rows -> sums, all-reduced underneath the table backward
    if (S > 0 && s->n_part > 0) {
        STEP_TRY(ngp_reduce_partials2(b.partials, c.n_density, b.partials + (size_t)s->n_part * c.n_density, c.n_rgb, s->n_part, x.small, main_stream));
        if (s->scaler_on) {
            // (flag sets alternate per tail: set (tails & 1) is this step's, the other still holds the previous step's until this step's check clears it)
            const int32_t* last_shard = (x.mode >= 1 && s->tails > 0) ? x.flags + 8 * (1 - (int)(s->tails & 1)) + 4 : nullptr;
            hipLaunchKernelGGL(poison_small_kernel, dim3(1), dim3(1), 0, main, x.small, s->guard_armed ? s->guard + s->guard_parity : nullptr, last_shard);
            STEP_TRY(NGP_LAUNCH_RESULT());
        }
        s->guard_armed = false;
    } else {
        // this rank's batch had no samples: it joins every collective with zeros (DDP semantics: every rank joins every all-reduce)
        STEP_HIP(hipMemsetAsync(x.small, 0, sizeof(float) * (size_t)(c.n_density + c.n_rgb), main));
        STEP_HIP(hipMemsetAsync(x.grad_padded, 0, sizeof(ngp_half) * (size_t)padded, main));
    }
    STEP_HIP(hipEventRecord(s->ev_small, main));
    STEP_HIP(hipStreamWaitEvent(cs, s->ev_small, 0));
    STEP_TRY(ngp_comm_all_reduce(comm, x.small, c.n_density + c.n_rgb, NGP_COMM_F32, (ngp_stream_t)cs));
    // (2) table backward in launch groups; a chunk of the gradient leaves behind the group that completes it
    int next_chunk = 0;
    auto hand_over = [&](int64_t values_done) -> int {
        while (next_chunk < x.n_chunks && std::min<int64_t>((int64_t)(next_chunk + 1) * C, c.n_grid) <= values_done) {
            STEP_HIP(hipEventRecord(s->ev_chunk[next_chunk], main));
            STEP_HIP(hipStreamWaitEvent(cs, s->ev_chunk[next_chunk], 0));
            if (next_chunk == 0 && timing) STEP_HIP(hipEventRecord(s->ev_x0, cs));
            STEP_TRY(exchange_chunk(s, next_chunk, cs));
            ++next_chunk;
        }
        return 0;
    };
    if (S > 0) {
        if (s->binned) {
            for (int g = 0; g < x.n_groups; ++g) {
                STEP_TRY(table_backward_group(s, x.n_groups, g, nullptr, main_stream));
                if (g + 1 < x.n_groups) STEP_TRY(hand_over(s->group_end[g]));
            }
        } else {
            STEP_TRY(ngp_hashgrid_bwd_sliced(b.xyzs, c.xyz_min, c.xyz_max, b.dfeats, &c.meta, S, b.active, b.n_active, c.grid_grad16, main_stream));
        }
    }
    mark(s, 7, main);
    if (timing) STEP_HIP(hipEventRecord(s->ev_m, main));
    STEP_TRY(hand_over(c.n_grid));
    STEP_TRY(march_next_if_at(s, AT_HASHGRID_BWD));
    // (3) non-finite checks on the REDUCED buffers (alternating flag sets: each launch clears the other set's flag), then Adam
    const int k = (int)(s->tails & 1);
    ++s->tails;
    int32_t* mlp_cur = x.flags + 8 * k; int32_t* mlp_nxt = x.flags + 8 * (1 - k);
    int32_t* grid_cur = x.flags + 8 * k + 4; int32_t* grid_nxt = x.flags + 8 * (1 - k) + 4;
    const int n_small = c.n_density + c.n_rgb;
    ngp_stream_t cst = (ngp_stream_t)cs;
    if (x.mode >= 1) {
        STEP_TRY(ngp_found_inf2(x.small, 1, n_small, nullptr, 0, 0, mlp_cur, mlp_nxt, cst));
        STEP_TRY(ngp_found_inf2(x.shard16, 0, (int64_t)x.n_chunks * x.piece, nullptr, 0, 0, grid_cur, grid_nxt, cst));
        STEP_TRY(hand_over_scaler(s));
        {
            const int rc_adam = ngp_adam_step_field_pieces(c.enc_param + c.n_density, c.enc_half + c.n_density, x.shard16, c.enc_m + c.n_density, c.enc_v + c.n_density,
                                            c.n_grid, x.piece, x.n_chunks, comm->world, comm->rank,
                                            c.enc_param, c.enc_half, x.small, c.enc_m, c.enc_v, c.n_density,
                                            c.rgb_param, c.rgb_half, x.small + c.n_density, c.rgb_m, c.rgb_v, c.n_rgb,
                                            1, lr, c.beta1, c.beta2, c.eps, c.weight_decay, step, grad_scale, mlp_cur, grid_cur, x.step_state, cst);
            withdraw_scaler();                       // (consumed by a launch that went out; withdrawn if validation refused it)
            if (rc_adam) return rc_adam;
        }
        // the updated f16 table: every rank's pieces to every rank, in place (one RCCL group: one launch for all chunks)
        if (x.mode == 2) {
            STEP_TRY(ngp_comm_all_gather_direct(comm, x.table_padded, x.piece, NGP_COMM_F16, cst));
        } else {
        if (x.n_chunks > 1) STEP_TRY(ngp_comm_group_begin());
        int rc = 0;
        for (int ch = 0; ch < x.n_chunks && rc == 0; ++ch) {
            ngp_half* t = x.table_padded + (size_t)ch * C;
            rc = ngp_comm_all_gather(comm, t + (size_t)comm->rank * x.piece, t, x.piece, NGP_COMM_F16, cst);
        }
        if (x.n_chunks > 1) { const int rc2 = ngp_comm_group_end(); if (rc == 0) rc = rc2; }
        STEP_TRY(rc);
        }
    } else {
        // one flag for everything (all ranks hold the same sums): GradScaler's whole-step decision
        STEP_TRY(ngp_found_inf2(x.grad_padded, 0, padded, x.small, 1, n_small, mlp_cur, mlp_nxt, cst));
        STEP_TRY(hand_over_scaler(s));
        {
            const int rc_adam = ngp_adam_step_field(c.enc_param + c.n_density, c.enc_half + c.n_density, c.grid_grad16, c.enc_m + c.n_density, c.enc_v + c.n_density, c.n_grid,
                                     c.enc_param, c.enc_half, x.small, c.enc_m, c.enc_v, c.n_density,
                                     c.rgb_param, c.rgb_half, x.small + c.n_density, c.rgb_m, c.rgb_v, c.n_rgb,
                                     1, lr, c.beta1, c.beta2, c.eps, c.weight_decay, step, grad_scale, 0, mlp_cur, x.step_state, cst);
            withdraw_scaler();                       // (consumed by a launch that went out; withdrawn if validation refused it)
            if (rc_adam) return rc_adam;
        }
    }
    if (timing) { STEP_HIP(hipEventRecord(s->ev_x1, cs)); s->x_times_set = true; }
    // (4) the one wait of the main stream: the next forward, the occupancy update and the next tail's memsets read / write what
    //     the communicator's stream has just produced / consumed
    STEP_HIP(hipEventRecord(s->ev_done, cs));
    STEP_HIP(hipStreamWaitEvent(main, s->ev_done, 0));
    mark(s, 8, main);
    STEP_TRY(march_next_if_at(s, AT_ADAM));
    return 0;
}

int ngp_stepper_exchange_times(ngp_stepper* s, float* exchange_ms, float* exposed_ms) {
    if (!s || !exchange_ms || !exposed_ms) return NGP_EINVAL;
    *exchange_ms = *exposed_ms = -1.0f;
    if (!s->comm || !s->x_times_set) return 0;
    STEP_HIP(hipEventSynchronize(s->ev_x1));
    STEP_HIP(hipEventElapsedTime(exchange_ms, s->ev_x0, s->ev_x1));
    STEP_HIP(hipEventElapsedTime(exposed_ms, s->ev_m, s->ev_x1));
    return 0;
}

int ngp_stepper_host_times(ngp_stepper* s, double* wait_s, double* enqueue_s, long long* n_steps, int reset) {
    if (!s || !wait_s || !enqueue_s || !n_steps) return NGP_EINVAL;
    *wait_s = s->t_wait; *enqueue_s = s->t_enqueue; *n_steps = s->n_fronts;
    if (reset) { s->t_wait = s->t_enqueue = 0.0; s->n_fronts = 0; }
    return 0;
}

int ngp_stepper_set_loss_scaler(ngp_stepper* s, float init_scale, float growth_factor, float backoff_factor, int32_t growth_interval,
                                ngp_stream_t main_stream) {
    if (!s) return NGP_EINVAL;
    if (!(init_scale > 0.f)) { s->scaler_on = false; return 0; }
    if (!(growth_factor >= 1.0f) || !(backoff_factor > 0.f && backoff_factor <= 1.0f) || growth_interval < 1) return NGP_EINVAL;
    const struct { float scale[2]; int32_t tracker[2]; } init = {{init_scale, init_scale}, {0, 0}};
    hipStream_t st = ngp_stream(main_stream);
    STEP_HIP(hipMemcpyAsync(s->scaler_state, &init, sizeof(init), hipMemcpyHostToDevice, st));
    STEP_HIP(hipStreamSynchronize(st));                  // (`init` lives on this frame)
    s->scaler_growth = growth_factor; s->scaler_backoff = backoff_factor; s->scaler_interval = growth_interval;
    s->scaler_slot = 0;
    s->scaler_on = true;
    return 0;
}

int ngp_stepper_loss_scale(ngp_stepper* s, float* scale, int32_t* growth_tracker, ngp_stream_t main_stream) {
    if (!s || !scale) return NGP_EINVAL;
    *scale = 0.f;
    if (growth_tracker) *growth_tracker = 0;
    if (!s->scaler_on) return 0;
    struct { float scale[2]; int32_t tracker[2]; } st;
    hipStream_t q = ngp_stream(main_stream);
    STEP_HIP(hipStreamSynchronize(q));
    if (s->comm) STEP_HIP(hipStreamSynchronize(s->comm->stream));          // (the exchange's optimizer launch runs on the communicator's stream)
    STEP_HIP(hipMemcpy(&st, s->scaler_state, sizeof(st), hipMemcpyDeviceToHost));
    *scale = st.scale[s->scaler_slot];
    if (growth_tracker) *growth_tracker = st.tracker[s->scaler_slot];
    return 0;
}

int ngp_stepper_timing(ngp_stepper* s, int enable) {
    if (!s) return NGP_EINVAL;
    s->timing = enable ? 1 : 0;
    return 0;
}

int ngp_stepper_stage_times(ngp_stepper* s, float* ms) {
    if (!s || !ms) return NGP_EINVAL;
    for (int i = 0; i < NGP_STEPPER_STAGES; ++i) ms[i] = -1.0f;
    int prev = -1;
    for (int i = 0; i < N_MARKS; ++i) {
        if (!s->mark_set[i]) continue;
        STEP_HIP(hipEventSynchronize(s->mark[i]));
        if (prev >= 0 && i >= 1) {
            float t = 0.f;
            STEP_HIP(hipEventElapsedTime(&t, s->mark[prev], s->mark[i]));
            ms[i - 1] = t;
        }
        prev = i;
    }
    const int k = s->last_set;
    if (s->march_t_set[k]) {
        STEP_HIP(hipEventSynchronize(s->march_t[k][1]));
        float t = 0.f;
        STEP_HIP(hipEventElapsedTime(&t, s->march_t[k][0], s->march_t[k][1]));
        ms[8] = t;
    }
    return 0;
}

#pragma GCC visibility pop
}  // extern "C"
