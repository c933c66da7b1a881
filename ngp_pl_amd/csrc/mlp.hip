// 64-wide fully fused MLPs (+ SH degree-4 encoding, TruncExp) on gfx950 matrix cores.
//
// Replaces tiny-cuda-nn's FullyFusedMLP / SphericalHarmonics as the reference configures them
// (/root/reference/models/networks.py:49-77): no biases, ReLU hidden activations, weights
// (out,in) row-major, f16 storage, output layer padded to 16 rows, f16 activations between
// layers.  tiny-cuda-nn is an un-vendored, unpinned dependency of the reference; semantics are
// restated from its published source and checked against
// oracle/tcnn_oracle.py.
//
// MI355X mapping.  One wave64 owns a tile of 32 samples and runs the whole network on
// v_mfma_f32_32x32x16_f16 with the operands SWAPPED: it computes Y^T = W * X^T, so the MFMA
// "N" index (lane & 31) is the SAMPLE and the "M" index is the output neuron.  The D fragment
// of layer n (neuron rows spread over registers, one sample per lane) is then already a valid
// B fragment for layer n+1 once the K slots are renamed -- and a K renaming is free, it only
// changes which weight columns the A fragment loads.  Activations therefore never leave
// registers in forward/dgrad; no LDS round trip, no barrier (tiny-cuda-nn goes through shared
// memory between layers because wmma keeps samples on M).  Only the weight gradient needs the
// sample on K, i.e. both operands transposed: each wave parks its tile's activations and gradients
// in a private LDS image in the layout it holds them in and reads them back through gfx950's
// transposing LDS read (ds_read_b64_tr_b16; see TrAddr below).
//
// K-slot renaming: MFMA element e (0..7) of lane-half hh (lane>>5) in K-chunk c is
//   natural order : unit 16c + 8hh + e                      (operand loaded from memory)
//   D order       : unit 16c + 4hh + (e&3) + 8(e>>2)        (operand is a previous D fragment;
//                   D register r of lane-half hh holds row (r&3) + 8(r>>2) + 4hh)
// Weights live in LDS (row pitch padded by 16 B -> conflict-free ds_read_b128/b64).
#include "ngp_common.h"

namespace {

typedef _Float16 h1;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HID = 64;          // neurons
constexpr int PAD = 8;           // halves of row padding in LDS
constexpr int WAVES = 4;         // waves per workgroup
// backward: waves per workgroup, one workgroup per CU.  Two hidden layers: 128 accumulator registers and 20 KB of LDS image per
// wave -> 4 waves, one per SIMD.  One hidden layer (the density net): 64 accumulators, 228 registers in all, 12 KB of image ->
// 8 waves, two per SIMD, which hide each other's LDS / MFMA latencies.
#ifndef NGP_MLP_BWD_WAVES_1HIDDEN
#define NGP_MLP_BWD_WAVES_1HIDDEN 8
#endif
// Two hidden layers at 8 waves (round 3, NGP_MLP_BWD_WAVES_2HIDDEN=4 builds the round-2 kernel): the operand image shrinks to 7
// blocks per wave (14 KB instead of 20: the first layer's operands go where the upper layers' have been read), the dW reduction
// takes the 64-row layers in two 32-row passes (8 x 8 KB of slabs instead of 8 x 16 KB), and the register allocation is held to
// 256 by the 512-thread launch bound.
#ifndef NGP_MLP_BWD_WAVES_2HIDDEN
#define NGP_MLP_BWD_WAVES_2HIDDEN 4
#endif
template <int N_IN, int N_HIDDEN>
constexpr int bwd_waves() { return N_IN > 32 ? 4 : (N_HIDDEN == 1 ? NGP_MLP_BWD_WAVES_1HIDDEN : NGP_MLP_BWD_WAVES_2HIDDEN); }
template <int N_IN, int N_HIDDEN>
constexpr bool bwd_staged() { return N_HIDDEN == 2 && bwd_waves<N_IN, N_HIDDEN>() > 4; }
constexpr int TILE = 32;         // samples per wave tile

__device__ __forceinline__ f32x16 mfma(half8_t a, half8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() { f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0.f; return z; }

// ---- A fragments from LDS weights (row-major, pitch ld halves) ----
__device__ __forceinline__ half8_t ldsA_nat(const h1* W, int ld, int row, bool valid, int c, int hh) {
    half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return valid ? *reinterpret_cast<const half8_t*>(W + row * ld + 16 * c + 8 * hh) : z;
}
__device__ __forceinline__ half8_t ldsA_dl(const h1* W, int ld, int row, bool valid, int c, int hh) {
    half8_t r = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        const half4_t lo = *reinterpret_cast<const half4_t*>(W + row * ld + 16 * c + 4 * hh);
        const half4_t hi = *reinterpret_cast<const half4_t*>(W + row * ld + 16 * c + 4 * hh + 8);
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    }
    return r;
}

// D fragment (f32) -> B fragment (f16) for chunk half c2 (registers 8*c2 .. 8*c2+7).  ReLU is applied to the PACKED halves as
// an INTEGER max of the bit patterns against 0 (4 v_pk_max_i16 where an f32 max of an MFMA result cost 2 x 16 VALU ops: the
// compiler canonicalises first): positive halves are positive int16 and stay, everything with the sign bit -- negative values
// and the -0 a tiny negative pre-activation rounds to -- becomes +0, exactly what (half)fmaxf(v, 0) gives.
template <bool RELU>
__device__ __forceinline__ half8_t d_to_b(const f32x16& d, int c2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    half8_t b;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {                      // <2 x float> -> <2 x half>: one v_cvt_pk_f16_f32 (round to nearest even)
        const f32x2 p = {d[8 * c2 + e], d[8 * c2 + e + 1]};
        const half2_t q = __builtin_convertvector(p, half2_t);
        b[e] = q[0]; b[e + 1] = q[1];
    }
    if (RELU) {
        typedef short short8_t __attribute__((ext_vector_type(8)));
        const short8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        b = __builtin_bit_cast(half8_t, __builtin_elementwise_max(__builtin_bit_cast(short8_t, b), z));
    }
    return b;
}

// ReLU backward on packed halves: d where h > 0, else 0.  h is a d_to_b<true>() result (positive or +0): per 16-bit half
// min(bits, 1) is 0/1, its negation the AND mask -- 3 packed integer ops per pair instead of a compare + select per element.
__device__ __forceinline__ half8_t relu_bwd(half8_t d, half8_t h) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t hb = __builtin_bit_cast(u32x4_t, h);
    u32x4_t db = __builtin_bit_cast(u32x4_t, d);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned int m;
        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(hb[k]));
        asm("v_pk_sub_u16 %0, 0, %1 op_sel_hi:[0,1]" : "=v"(m) : "v"(m));
        db[k] &= m;
    }
    return __builtin_bit_cast(half8_t, db);
}

// Stage a (rows, cols) row-major f16 matrix from global into LDS with padded pitch, optionally
// transposed (LDS gets (cols, rows)).  Whole workgroup participates; 16-byte global loads (cols is a multiple of 8, blobs are
// 16 B aligned).  Transposed: consecutive lanes take consecutive ROWS, so the eight 2-byte stores of a lane's chunk land on
// consecutive halves across the wave (lanes on consecutive columns put 16 lanes on one bank).
__device__ __forceinline__ void stage_weights(const h1* __restrict__ g, h1* lds, int rows, int cols, bool transpose) {
    const int n8 = rows * cols / 8, chunks = cols / 8;
    for (int t = threadIdx.x; t < n8; t += blockDim.x) {
        if (transpose) {
            const int r = t % rows, c = 8 * (t / rows);
            const half8_t v = *reinterpret_cast<const half8_t*>(g + r * cols + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[(c + e) * (rows + PAD) + r] = v[e];
        } else {
            const int r = t / chunks, c = 8 * (t - r * chunks);
            *reinterpret_cast<half8_t*>(lds + r * (cols + PAD) + c) = *reinterpret_cast<const half8_t*>(g + 8 * t);
        }
    }
}

// ---- SH degree 4 (tiny-cuda-nn spherical_harmonics.h, constants as SURVEY.md section 8a) ----
template <class O>
__device__ __forceinline__ void sh4(float x, float y, float z, O& o) {
    // no mul+add contraction: every kernel that evaluates the basis (forward, both backwards, ngp_sh4_fwd) gets the SAME sixteen
    // f32 values -- the plain products and sums the fp32 restatement computes -- instead of whatever mix of FMAs each inlined copy
    // was given (round 6: the one-launch backward differed from the two-launch one by an f16 ulp of one coefficient in 2 % of tiles)
#pragma clang fp contract(off)
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// SH(d/|d|) as the B fragment of input chunk 0 (natural K order: lane-half hh holds coefficients 8hh .. 8hh+7).  The coefficients
// live in a VECTOR value: as a float[16] indexed by 8*hh + e they went through scratch memory (80 bytes, a store + load round trip per tile).
__device__ __forceinline__ half8_t sh4_fragment(float dx, float dy, float dz, int hh) {
#pragma clang fp contract(off)
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    f32x16 sh;
    sh4(dx * inv, dy * inv, dz * inv, sh);
    half8_t b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float lo = sh[e], hi = sh[8 + e];
        b[e] = (h1)(hh ? hi : lo);
    }
    return b;
}

enum InMode { IN_ROWMAJOR = 0, IN_LEVELMAJOR = 1, IN_SH_H = 2 };
enum OutMode { OUT_PLAIN = 0, OUT_DENSITY = 1, OUT_RGB = 2 };

struct MlpIO {
    const h1* in;          // IN_ROWMAJOR: (S,N_IN); IN_LEVELMAJOR: [16][S] half2; IN_SH_H: h (S,16)
    const float* dirs;     // IN_SH_H: (S,3) un-normalised
    h1* out16;             // (S, out_ld) f16, may be null
    int out_ld;            // row length of out16 (n_out for tcnn.Network, 16 for h)
    int n_out;             // columns actually written
    float* sigmas;         // OUT_DENSITY
    float* rgbs;           // OUT_RGB (S,3)
    int out_act;           // OUT_PLAIN: 0 none, 1 sigmoid
    const int32_t* n_dev;  // optional device-side sample count (forward only; the launch covers an upper bound)
    const int32_t* scatter;// OUT_DENSITY, optional: sigma of sample s goes to sigmas[scatter[s]]
};

// Load the B fragments (natural K order) of the network input for this lane's sample.
template <int N_IN, int IN_MODE>
__device__ __forceinline__ void load_input(const MlpIO& io, long long s, bool valid, int n_samples, int hh,
                                           half8_t (&b)[N_IN / 16]) {
    const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < N_IN / 16; ++c) b[c] = z;
    if (!valid) return;
    if (IN_MODE == IN_ROWMAJOR) {
#pragma unroll
        for (int c = 0; c < N_IN / 16; ++c)
            b[c] = *reinterpret_cast<const half8_t*>(io.in + s * N_IN + 16 * c + 8 * hh);
    } else if (IN_MODE == IN_LEVELMAJOR) {
        const half2_t* f = reinterpret_cast<const half2_t*>(io.in);
#pragma unroll
        for (int c = 0; c < N_IN / 16; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half2_t v = __builtin_nontemporal_load(f + (size_t)(8 * c + 4 * hh + q) * n_samples + s);
                b[c][2 * q] = v[0]; b[c][2 * q + 1] = v[1];
            }
    } else {  // IN_SH_H: chunk 0 = SH(d/|d|), chunk 1 = h
        b[0] = sh4_fragment(io.dirs[3 * s], io.dirs[3 * s + 1], io.dirs[3 * s + 2], hh);
        b[1] = *reinterpret_cast<const half8_t*>(io.in + s * 16 + 8 * hh);
    }
}

// Forward through the hidden stack.  LDS weights: W0 (64, N_IN), W1 (64,64) [if N_HIDDEN==2],
// Wo (16,64); pitches padded.  Leaves hidden activations (post-ReLU, f16 B fragments) in hb*.
template <int N_IN, int N_HIDDEN>
struct LdsW {
    static constexpr int LD0 = N_IN + PAD, LDH = HID + PAD;
    static constexpr int OFF_W0 = 0;
    static constexpr int OFF_W1 = OFF_W0 + HID * LD0;
    static constexpr int OFF_WO = OFF_W1 + (N_HIDDEN == 2 ? HID * LDH : 0);
    static constexpr int SIZE = OFF_WO + 16 * LDH;          // halves
    // global blob offsets (tight)
    static constexpr int G_W1 = HID * N_IN;
    static constexpr int G_WO = G_W1 + (N_HIDDEN == 2 ? HID * HID : 0);
    static constexpr int G_SIZE = G_WO + 16 * HID;
};

template <int N_IN, int N_HIDDEN>
__device__ __forceinline__ void stage_fwd_weights(const h1* __restrict__ w, h1* lds) {
    using L = LdsW<N_IN, N_HIDDEN>;
    stage_weights(w, lds + L::OFF_W0, HID, N_IN, false);
    if (N_HIDDEN == 2) stage_weights(w + L::G_W1, lds + L::OFF_W1, HID, HID, false);
    stage_weights(w + L::G_WO, lds + L::OFF_WO, 16, HID, false);
}

// The same image from ONE pass over the (tight) blob: all of a thread's 16-byte loads issued first -- compile-time trip count, fully
// coalesced -- then the stores into the padded rows: one global round trip at the head of the kernel instead of one per matrix and
// loop trip (three to six dependent ones: 3-4 us of the 26 us field forward).  THREADS = the workgroup's size.
#ifndef NGP_MLP_FWD_STAGE_ONCE
#define NGP_MLP_FWD_STAGE_ONCE 1          // 0: matrix by matrix (A/B builds)
#endif
template <int N_IN, int N_HIDDEN, int THREADS>
__device__ __forceinline__ void stage_fwd_weights_once(const h1* __restrict__ w, h1* lds) {
    using L = LdsW<N_IN, N_HIDDEN>;
    if (!NGP_MLP_FWD_STAGE_ONCE) { stage_fwd_weights<N_IN, N_HIDDEN>(w, lds); return; }
    constexpr int C0 = HID * N_IN / 8, C1 = (N_HIDDEN == 2 ? HID * HID / 8 : 0), CO = 16 * HID / 8, CT = C0 + C1 + CO;   // 16-byte chunks
    constexpr int PER = (CT + THREADS - 1) / THREADS;
    half8_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = (int)threadIdx.x + k * THREADS;
        v[k] = *reinterpret_cast<const half8_t*>(w + 8 * (t < CT ? t : 0));
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = (int)threadIdx.x + k * THREADS;
        if (t >= CT) continue;
        int cols, tm, off;
        if (t < C0) { cols = N_IN; tm = t; off = L::OFF_W0; }
        else if (t < C0 + C1) { cols = HID; tm = t - C0; off = L::OFF_W1; }
        else { cols = HID; tm = t - C0 - C1; off = L::OFF_WO; }
        const int chunks = cols / 8, r = tm / chunks, c = 8 * (tm - r * chunks);
        *reinterpret_cast<half8_t*>(lds + off + r * (cols + PAD) + c) = v[k];
    }
}

// Backward kernels: the forward-layout copy AND the transposed copy of every matrix from ONE pass over the blob -- all of a thread's
// 16-byte global loads issued first (compile-time trip count), then the stores: one global round trip in the kernel's prologue instead of the
// seven dependent ones of stage_fwd_weights() + three stage_weights(transpose) calls (the prologue was 15 % / 29 % of the colour / density
// kernel at the bench's operating point, tools/bench_mlp.py with -DNGP_MLP_TIMING).  Same LDS images bit for bit.
#ifndef NGP_MLP_STAGE_ONCE
#define NGP_MLP_STAGE_ONCE 1
#endif
template <int N_IN, int N_HIDDEN, int THREADS>
__device__ __forceinline__ void stage_bwd_weights(const h1* __restrict__ w, h1* lds, int off_t0, int off_t1, int off_to) {
    using L = LdsW<N_IN, N_HIDDEN>;
    constexpr int C0 = HID * N_IN / 8, C1 = (N_HIDDEN == 2 ? HID * HID / 8 : 0), CO = 16 * HID / 8, CT = C0 + C1 + CO;   // 16-byte chunks
    constexpr int PER = (CT + THREADS - 1) / THREADS;
    half8_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = (int)threadIdx.x + k * THREADS;
        // chunk t of the blob: matrix m (rows x cols), chunk tm inside it; lanes on consecutive ROWS (see stage_weights)
        int rows, cols, tm, gbase;
        if (t < C0) { rows = HID; cols = N_IN; tm = t; gbase = 0; }
        else if (t < C0 + C1) { rows = HID; cols = HID; tm = t - C0; gbase = L::G_W1; }
        else { rows = 16; cols = HID; tm = t - C0 - C1; gbase = L::G_WO; }
        const int r = tm % rows, c = 8 * (tm / rows);
        const int tc = t < CT ? gbase + r * cols + c : 0;
        v[k] = *reinterpret_cast<const half8_t*>(w + tc);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = (int)threadIdx.x + k * THREADS;
        if (t >= CT) continue;
        int rows, cols, tm, off_f, off_t;
        if (t < C0) { rows = HID; cols = N_IN; tm = t; off_f = L::OFF_W0; off_t = off_t0; }
        else if (t < C0 + C1) { rows = HID; cols = HID; tm = t - C0; off_f = L::OFF_W1; off_t = off_t1; }
        else { rows = 16; cols = HID; tm = t - C0 - C1; off_f = L::OFF_WO; off_t = off_to; }
        const int r = tm % rows, c = 8 * (tm / rows);
        *reinterpret_cast<half8_t*>(lds + off_f + r * (cols + PAD) + c) = v[k];
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[off_t + (c + e) * (rows + PAD) + r] = v[k][e];
    }
}

// hidden layer from natural-order input fragments: out tiles m=0,1 (64 neurons)
template <int N_IN>
__device__ __forceinline__ void layer_in(const h1* W, const half8_t (&b)[N_IN / 16], int i, int hh, f32x16 (&acc)[2]) {
    acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
    for (int c = 0; c < N_IN / 16; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m)
            acc[m] = mfma(ldsA_nat(W, N_IN + PAD, 32 * m + i, true, c, hh), b[c], acc[m]);
}
// 64 -> (32*M_TILES rows, first n_rows valid) from D-order fragments hb[4]
template <int M_TILES>
__device__ __forceinline__ void layer_hid(const h1* W, int ld, int n_rows, const half8_t (&hb)[4], int i, int hh,
                                          f32x16 (&acc)[M_TILES]) {
#pragma unroll
    for (int m = 0; m < M_TILES; ++m) acc[m] = zero16();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < M_TILES; ++m)
            acc[m] = mfma(ldsA_dl(W, ld, 32 * m + i, (32 * m + i) < n_rows, c, hh), hb[c], acc[m]);
}
template <bool RELU>
__device__ __forceinline__ void acc_to_frag(const f32x16 (&acc)[2], half8_t (&hb)[4]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) hb[2 * m + c2] = d_to_b<RELU>(acc[m], c2);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------
// forward kernel
// ------------------------------------------------------------------------------------------
template <int N_IN, int N_HIDDEN, int IN_MODE, int OUT_MODE>
__global__ void __launch_bounds__(64 * WAVES)
mlp_fwd_kernel(MlpIO io, const h1* __restrict__ weights, int n_samples) {
    using L = LdsW<N_IN, N_HIDDEN>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h1* lds = reinterpret_cast<h1*>(smem_raw);
    if (io.n_dev != nullptr) {
        n_samples = min(*io.n_dev, n_samples);
        if ((long long)blockIdx.x * WAVES * TILE >= n_samples) return;
    }
    stage_fwd_weights_once<N_IN, N_HIDDEN, 64 * WAVES>(weights, lds);
    __syncthreads();

    const int lane = threadIdx.x & 63, i = lane & 31, hh = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const int n_tiles = (n_samples + TILE - 1) / TILE;
    for (int tile = blockIdx.x * WAVES + wave; tile < n_tiles; tile += gridDim.x * WAVES) {
        const long long s = (long long)tile * TILE + i;
        const bool valid = s < n_samples;
        half8_t xb[N_IN / 16];
        load_input<N_IN, IN_MODE>(io, s, valid, n_samples, hh, xb);
        f32x16 acc[2];
        half8_t hb[4];
        layer_in<N_IN>(lds + L::OFF_W0, xb, i, hh, acc);
        acc_to_frag<true>(acc, hb);
        if (N_HIDDEN == 2) {
            layer_hid<2>(lds + L::OFF_W1, L::LDH, 64, hb, i, hh, acc);
            acc_to_frag<true>(acc, hb);
        }
        f32x16 o[1];
        layer_hid<1>(lds + L::OFF_WO, L::LDH, 16, hb, i, hh, o);
        if (!valid) continue;
        // o[0][r], r<8: output unit 4hh + (r&3) + 8(r>>2)
        if (OUT_MODE == OUT_DENSITY) {
            half4_t lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = (h1)o[0][e]; hi[e] = (h1)o[0][4 + e]; }
            if (io.out16) {
                *reinterpret_cast<half4_t*>(io.out16 + s * 16 + 4 * hh) = lo;
                *reinterpret_cast<half4_t*>(io.out16 + s * 16 + 8 + 4 * hh) = hi;
            }
            if (hh == 0) {
                const float sg = __expf((float)lo[0]);                 // TruncExp fwd on the f16 h[0] (networks.py:105)
                // scatter (occupancy update): a cell drawn twice keeps the LARGER of its densities -- sigma >= 0, so the order of the
                // bit patterns is the order of the values and an integer max decides, whatever order the stores arrive in.  (torch's
                // indexed assignment, networks.py:256-258, keeps "one of them", unspecified on CUDA: with a plain store here two runs of
                // the same training diverged from the first sampled update on, step 256.)  The target is zero-filled by the caller.
                if (io.scatter) atomicMax(reinterpret_cast<unsigned int*>(io.sigmas + (long long)io.scatter[s]), __float_as_uint(sg));
                else io.sigmas[s] = sg;
            }
        } else if (OUT_MODE == OUT_RGB) {
            if (hh == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) io.rgbs[3 * s + c] = (float)(h1)sigmoidf(o[0][c]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int u = 4 * hh + (r & 3) + 8 * (r >> 2);
                if (u < io.n_out) {
                    float v = o[0][r];
                    if (io.out_act == 1) v = sigmoidf(v);
                    io.out16[s * io.out_ld + u] = (h1)v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// fused field forward: density net -> (sigma, h) -> rgb net in ONE pass.  h never leaves
// registers: the density net's D fragment (16 outputs, D order) is chunk 1 of the colour net's
// input (chunk 0 = SH of the direction), consumed through D-order A fragments of W0.
// ------------------------------------------------------------------------------------------
struct FieldIO {
    const h1* feats;       // [16][S] half2
    const float* dirs;     // (S,3)
    float* sigmas;         // (S)
    float* rgbs;           // (S,3)
    h1* h_out;             // (S,16), may be null (inference)
    const int32_t* n_dev;  // optional device-side sample count
    const int32_t* list;   // LIST instantiation: work item j handles sample list[j]; n_samples is then the list length (bound) and
    int stride;            //                     stride the level stride of feats / the bound of the ids
};

template <bool LIST>
__global__ void __launch_bounds__(64 * WAVES)
field_fwd_kernel(FieldIO io, const h1* __restrict__ density_w, const h1* __restrict__ rgb_w, int n_samples) {
    using LD = LdsW<32, 1>;
    using LR = LdsW<32, 2>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h1* ldsd = reinterpret_cast<h1*>(smem_raw);
    h1* ldsr = ldsd + LD::SIZE;
    if (io.n_dev != nullptr) {
        n_samples = min(*io.n_dev, n_samples);
        if ((long long)blockIdx.x * WAVES * TILE >= n_samples) return;
    }
    stage_fwd_weights_once<32, 1, 64 * WAVES>(density_w, ldsd);
    stage_fwd_weights_once<32, 2, 64 * WAVES>(rgb_w, ldsr);
    __syncthreads();

    const int lane = threadIdx.x & 63, i = lane & 31, hh = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const int n_tiles = (n_samples + TILE - 1) / TILE;
    // one tile of raw inputs ahead: the loads of tile t+1 are in flight while tile t runs through the networks (two waves per
    // SIMD alone left every wave waiting a memory round trip per tile).  Straight-line loads from a clamped sample position.
    const half2_t* fp = reinterpret_cast<const half2_t*>(io.feats);
    const long long s_last = n_samples - 1;
    const long long stride = LIST ? (long long)io.stride : (long long)n_samples;
    half8_t x_nxt[2]; float d_nxt[3];
    long long id_nxt = 0, id_pre = 0;                 // LIST: sample id of the tile being fetched / of the one after it (two ahead)
    bool pad_nxt = false, pad_pre = false;            //       ... and whether that list entry is padding (-1): computed, never stored
    auto list_id = [&](int t) -> long long {
        const long long sj = (long long)t * TILE + i;
        const long long v = io.list[sj < s_last ? sj : s_last];
        pad_pre = v < 0 || v >= stride;
        return pad_pre ? 0 : v;
    };
    auto fetch = [&](int t, long long id) {
        const long long sj = (long long)t * TILE + i;
        const long long sc = LIST ? id : (sj < s_last ? sj : s_last);
        id_nxt = sc; pad_nxt = pad_pre;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half2_t v = __builtin_nontemporal_load(fp + (size_t)(8 * c + 4 * hh + q) * stride + sc);
                x_nxt[c][2 * q] = v[0]; x_nxt[c][2 * q + 1] = v[1];
            }
        d_nxt[0] = io.dirs[3 * sc]; d_nxt[1] = io.dirs[3 * sc + 1]; d_nxt[2] = io.dirs[3 * sc + 2];
    };
    const int tile_stride = gridDim.x * WAVES;
    const int t_first = blockIdx.x * WAVES + wave;
    fetch(t_first, LIST ? list_id(t_first) : 0);
    if (LIST) id_pre = list_id(t_first + tile_stride);
    for (int tile = t_first; tile < n_tiles; tile += tile_stride) {
        const long long sj = (long long)tile * TILE + i;
        const bool valid = sj < n_samples && !(LIST && pad_nxt);
        const long long s = LIST ? id_nxt : sj;          // where this lane's sample lives
        half8_t xb[2] = {x_nxt[0], x_nxt[1]};
        const float dx = d_nxt[0], dy = d_nxt[1], dz = d_nxt[2];
        fetch(tile + tile_stride, id_pre);
        if (LIST) id_pre = list_id(tile + 2 * tile_stride);
        f32x16 acc[2];
        half8_t hb[4];
        layer_in<32>(ldsd + LD::OFF_W0, xb, i, hh, acc);
        acc_to_frag<true>(acc, hb);
        f32x16 o[1];
        layer_hid<1>(ldsd + LD::OFF_WO, LD::LDH, 16, hb, i, hh, o);
        // h (f16): register e of lane-half hh is unit 4hh + (e&3) + 8(e>>2)
        half8_t hfrag;
#pragma unroll
        for (int e = 0; e < 8; ++e) hfrag[e] = (h1)o[0][e];
        // colour net input chunk 0: SH(d/|d|), natural K order
        const half8_t shfrag = sh4_fragment(dx, dy, dz, hh);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc[m] = mfma(ldsA_nat(ldsr + LR::OFF_W0, LR::LD0, 32 * m + i, true, 0, hh), shfrag, zero16());
            acc[m] = mfma(ldsA_dl(ldsr + LR::OFF_W0, LR::LD0, 32 * m + i, true, 1, hh), hfrag, acc[m]);
        }
        acc_to_frag<true>(acc, hb);
        layer_hid<2>(ldsr + LR::OFF_W1, LR::LDH, 64, hb, i, hh, acc);
        acc_to_frag<true>(acc, hb);
        f32x16 c[1];
        layer_hid<1>(ldsr + LR::OFF_WO, LR::LDH, 16, hb, i, hh, c);
        if (!valid) continue;
        if (io.h_out) {
            half4_t lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = hfrag[e]; hi[e] = hfrag[4 + e]; }
            *reinterpret_cast<half4_t*>(io.h_out + s * 16 + 4 * hh) = lo;
            *reinterpret_cast<half4_t*>(io.h_out + s * 16 + 8 + 4 * hh) = hi;
        }
        if (hh == 0) {
            io.sigmas[s] = __expf((float)hfrag[0]);          // TruncExp fwd on the f16 h[0] (networks.py:105)
#pragma unroll
            for (int k = 0; k < 3; ++k) io.rgbs[3 * s + k] = (float)(h1)sigmoidf(c[0][k]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward kernel: recompute forward, dgrad in registers, wgrad operands through a wave-private
// LDS image read back transposed, per-workgroup partial weight gradients.
// ------------------------------------------------------------------------------------------
struct MlpBwdIO {
    MlpIO fwd;               // inputs as in forward (outputs unused)
    const h1* dL_dout16;     // OUT_PLAIN / OUT_DENSITY: (S, dout_ld) f16 (already loss-scaled), may be null
    int dout_ld;
    const float* dL_dsigmas; // OUT_DENSITY: (S) f32 unscaled, may be null
    const float* dL_drgbs;   // OUT_RGB: (S,3) f32 unscaled
    float loss_scale;
    const float* loss_scale_dev;   // optional device-side factor on loss_scale (the native stepper's dynamic loss scale), read once per launch
    float din_limit;               // an input gradient beyond this magnitude raises `nonfinite` (0: 65504, the f16 range)
    h1* dL_din;              // IN_ROWMAJOR: (S,N_IN); IN_LEVELMAJOR: [16][S] half2; IN_SH_H: dh (S,16); may be null
    float* wgrad_partial;    // (gridDim.x, G_SIZE) f32
    // Optional compaction: only the samples active[0 .. *n_active) are processed.  Network inputs
    // and the f32 seeds are addressed by the sample id, the f16 gradient chain (dL_dout16 in,
    // dL_din out) by the compact position, so rgb-net -> density-net -> grid scatter stay dense.
    const int32_t* active;
    const int32_t* n_active;
    // Optional overflow guard (the native stepper's GradScaler: a step whose weight gradient is not finite is skipped by the
    // optimizer launch): a workgroup whose reduced dW holds an inf / NaN sets *nonfinite; workgroup 0 clears *nonfinite_clear (the
    // flag of the NEXT step: two flags alternate, so that no launch resets a word a later launch of the same step reads).
    int32_t* nonfinite;
    int32_t* nonfinite_clear;
};

// Lanes of one wave exchange data through LDS: DS operations of a wave execute in program
// order, so only the COMPILER has to be kept from moving reads above other lanes' writes.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Weight gradient operands.  dW = dY^T X needs the SAMPLE on the MFMA K index, i.e. both operands transposed with respect
// to how forward and dgrad hold them (lane = sample).  Every wave keeps its tile's activations and gradients in a private LDS
// image in the layout it owns them in -- blocks of [32 samples][32 units] f16, 64-byte rows, written with 8-byte stores (4
// consecutive units of one sample) -- and reads the operands back with gfx950's transposing LDS read (ds_read_b64_tr_b16: the 16
// lanes of a group hand in 16 x 4 halves and receive them transposed, 4 samples of one unit per lane).  38 stores + 40 reads per
// tile where element-wise transposition took 160 two-byte stores.
// The 8-byte pieces of a row are XOR-swizzled by (sample >> 1) & 7 so that the 32 lanes of a store spread over all banks;
// a read covers whole rows (4 samples x 64 B per half-wave), which no permutation inside a row can make conflict.
constexpr int BLK_BYTES = 32 * 32 * 2;
typedef __fp16 fp16x4_t __attribute__((__vector_size__(8)));

struct TrAddr {
    int wD[4];    // store offset of piece hh + 2k of this lane's sample row (D-order fragments: units 4hh+{0..3} (+8) of a 16-unit chunk)
    int wN[4];    // store offset of piece 2hh + 4(k>>1) + (k&1)            (natural-order fragments: units 8hh + {0..7} of a chunk)
    int r[2];     // read offset of this lane's 8 bytes for samples 8hh + 4t + {0..3} of K chunk 0 (chunk 1: + 16 rows)
};
__device__ __forceinline__ TrAddr tr_addr(int lane) {
    const int j = lane & 31, hh = lane >> 5, swz = (j >> 1) & 7;
    TrAddr a;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a.wD[k] = j * 64 + (((hh + 2 * k) ^ swz) << 3);
        a.wN[k] = j * 64 + (((2 * hh + 4 * (k >> 1) + (k & 1)) ^ swz) << 3);
    }
    const int q = lane & 15, blk16 = (lane >> 4) & 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int srow = 8 * hh + 4 * t + (q >> 2);
        a.r[t] = srow * 64 + (((4 * blk16 + (q & 3)) ^ ((srow >> 1) & 7)) << 3);
    }
    return a;
}
__device__ __forceinline__ void tr_st(char* p, const half8_t& b, int lo_or_hi) {
    half4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = b[4 * lo_or_hi + e];
    *reinterpret_cast<half4_t*>(p) = v;
}
// D-order fragments b[NCH] (unit 16c + 4hh + (e&3) + 8(e>>2)) -> blocks c>>1 of `arr`
template <int NCH>
__device__ __forceinline__ void tr_put_dl(char* arr, const half8_t (&b)[NCH], const TrAddr& a) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        tr_st(arr + (c >> 1) * BLK_BYTES + a.wD[2 * (c & 1)], b[c], 0);
        tr_st(arr + (c >> 1) * BLK_BYTES + a.wD[2 * (c & 1) + 1], b[c], 1);
    }
}
// natural-order fragments b[NCH] (unit 16c + 8hh + e)
template <int NCH>
__device__ __forceinline__ void tr_put_nat(char* arr, const half8_t (&b)[NCH], const TrAddr& a) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        tr_st(arr + (c >> 1) * BLK_BYTES + a.wN[2 * (c & 1)], b[c], 0);
        tr_st(arr + (c >> 1) * BLK_BYTES + a.wN[2 * (c & 1) + 1], b[c], 1);
    }
}
// MFMA operand of K chunk c (samples 16c .. 16c+15) for the 32 units of block `blk`: lane (i, hh) gets samples 16c + 8hh + e of unit i
__device__ __forceinline__ half8_t tr_get(const char* blk, int c, const TrAddr& a) {
    typedef __attribute__((address_space(3))) fp16x4_t* lds_p;
    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(blk + c * 1024 + a.r[0]));
    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(blk + c * 1024 + a.r[1]));
    half8_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (h1)lo[e]; v[4 + e] = (h1)hi[e]; }
    return v;
}
// dW(MT*32 x NT*32) += dY^T (rows) * X (cols) over the 32 samples of the tile
template <int MT, int NT>
__device__ __forceinline__ void wgrad_tile(const char* dy, const char* x, const TrAddr& ta, f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        half8_t a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = tr_get(dy + m * BLK_BYTES, c, ta);
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = tr_get(x + n * BLK_BYTES, c, ta);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = mfma(a[m], b[n], acc[m][n]);
    }
}
// Workgroup reduction of one layer's dW: every wave stores its accumulators into its own LDS slab (row-major (n_rows, n_cols)
// f32, lanes on consecutive columns: conflict-free), one barrier, then all threads add the slabs 16 bytes at a time and store
// the sums straight to the workgroup's partial row in global memory.  (Before: the waves took turns read-modify-writing one
// LDS copy, 4 serial rounds with barriers -- 11 000 cycles of a 70 000-cycle kernel.)
template <int MT, int NT, int BWD_WAVES, bool SPLIT>
__device__ __forceinline__ void wgrad_reduce_layer(float* slabs, int n_rows, int n_cols, int wave, int j, int hh,
                                                   const f32x16 (&acc)[MT][NT], float* __restrict__ out, float& nonfinite) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // SPLIT: one pass per 32-row block of the layer (slabs of 32 x n_cols floats), else the whole layer at once
    constexpr int PASSES = SPLIT ? MT : 1, MP = SPLIT ? 1 : MT;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int rows_here = SPLIT ? min(32, n_rows - 32 * p) : n_rows;
        const int n = rows_here * n_cols;                 // a multiple of 4 (n_cols is)
        float* mine = slabs + wave * n;
#pragma unroll
        for (int mm = 0; mm < MP; ++mm)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mm + (r & 3) + 8 * (r >> 2) + 4 * hh, col = 32 * nn + j;
                    if (row < rows_here && col < n_cols) mine[row * n_cols + col] = acc[SPLIT ? p : mm][nn][r];
                }
        __syncthreads();
        float* __restrict__ o = out + (SPLIT ? 32 * p * n_cols : 0);
        for (int t = threadIdx.x * 4; t < n; t += blockDim.x * 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(slabs + t);
#pragma unroll
            for (int w = 1; w < BWD_WAVES; ++w) v += *reinterpret_cast<const f32x4*>(slabs + w * n + t);
            *reinterpret_cast<f32x4*>(o + t) = v;
            const f32x4 d = v - v;                        // 0 for a finite sum, NaN for inf / NaN
            nonfinite += (d[0] + d[1]) + (d[2] + d[3]);
        }
        __syncthreads();                                  // the next pass / layer reuses the slabs
    }
}

// Everything a backward tile reads from global memory, as RAW register values: the loads of tile t+1 are issued before tile t
// is computed (the kernel runs ONE wave per SIMD -- 164 VGPRs, 100 KB of LDS -- so nothing else hides a global round trip;
// before this, every tile paid two dependent ones: active index -> inputs/seeds).
template <int N_IN>
struct BwdRaw {
    bool valid;
    half8_t in[N_IN / 16];   // IN_ROWMAJOR / IN_LEVELMAJOR: the input B fragments; IN_SH_H: in[1] = h, in[0] unused
    float dir[3];            // IN_SH_H
    float seed[3];           // OUT_RGB: dL_drgbs; OUT_DENSITY: seed[0] = dL_dsigmas
    h1 dout[8];              // OUT_PLAIN / OUT_DENSITY: dL_dout16 at units 4hh + (r&3) + 8(r>>2)
};

// Straight-line code: every lane loads, from positions the caller has clamped into range (lanes past the end re-read the last
// sample; their output gradient is forced to zero, which zeroes everything they contribute).  Loads under per-lane conditions
// come with zero-initialised destinations and branch merges, and the s_waitcnt pass answered those with vmcnt(0) in the middle
// of the prefetch.
template <int N_IN, int IN_MODE, int OUT_MODE>
__device__ __forceinline__ void bwd_fetch(const MlpBwdIO& io, long long j, long long s, bool valid, int n_samples, int hh, BwdRaw<N_IN>& r) {
    r.valid = valid;
    r.dir[0] = r.dir[1] = r.dir[2] = 1.0f;
    r.seed[0] = r.seed[1] = r.seed[2] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) r.dout[q] = (h1)0;
    if (IN_MODE == IN_ROWMAJOR) {
#pragma unroll
        for (int c = 0; c < N_IN / 16; ++c) r.in[c] = *reinterpret_cast<const half8_t*>(io.fwd.in + s * N_IN + 16 * c + 8 * hh);
    } else if (IN_MODE == IN_LEVELMAJOR) {
        const half2_t* f = reinterpret_cast<const half2_t*>(io.fwd.in);
#pragma unroll
        for (int c = 0; c < N_IN / 16; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half2_t v = __builtin_nontemporal_load(f + (size_t)(8 * c + 4 * hh + q) * n_samples + s);
                r.in[c][2 * q] = v[0]; r.in[c][2 * q + 1] = v[1];
            }
    } else {
        const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        r.in[0] = z;
        r.dir[0] = io.fwd.dirs[3 * s]; r.dir[1] = io.fwd.dirs[3 * s + 1]; r.dir[2] = io.fwd.dirs[3 * s + 2];
        r.in[1] = *reinterpret_cast<const half8_t*>(io.fwd.in + s * 16 + 8 * hh);
    }
    if (OUT_MODE == OUT_RGB) {
#pragma unroll
        for (int c = 0; c < 3; ++c) r.seed[c] = io.dL_drgbs[3 * s + c];
    } else {
        if (io.dL_dout16) {                                       // wave-uniform conditions from here on
            if (io.dout_ld == 16 && io.fwd.n_out == 16) {        // (S,16) rows: units 4hh+{0..3} and 8+4hh+{0..3} as two 8-byte loads
                const half4_t lo = *reinterpret_cast<const half4_t*>(io.dL_dout16 + j * 16 + 4 * hh);
                const half4_t hi = *reinterpret_cast<const half4_t*>(io.dL_dout16 + j * 16 + 8 + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) { r.dout[q] = lo[q]; r.dout[4 + q] = hi[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int u = 4 * hh + (q & 3) + 8 * (q >> 2);
                    const h1 v = io.dL_dout16[j * io.dout_ld + (u < io.fwd.n_out ? u : 0)];
                    r.dout[q] = u < io.fwd.n_out ? v : (h1)0;
                }
            }
        }
        if (OUT_MODE == OUT_DENSITY && io.dL_dsigmas) r.seed[0] = io.dL_dsigmas[s];
    }
}

// raw tile -> input B fragments (natural K order), as load_input() forms them
template <int N_IN, int IN_MODE>
__device__ __forceinline__ void bwd_input(const BwdRaw<N_IN>& r, int hh, half8_t (&b)[N_IN / 16]) {
#pragma unroll
    for (int c = 0; c < N_IN / 16; ++c) b[c] = r.in[c];
    if (IN_MODE == IN_SH_H) {
        const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        b[0] = z;
        b[0] = sh4_fragment(r.dir[0], r.dir[1], r.dir[2], hh);
    }
}

#ifdef NGP_MLP_TIMING
// A/B instrumentation (tools/build_variant.sh ... -DNGP_MLP_TIMING): core-clock cycles wave 0 of every workgroup spends per stage
__device__ unsigned long long g_mlp_t[16];
#define MLP_T(k) do { const long long now_ = clock64(); t_acc[k] += now_ - t_last; t_last = now_; } while (0)
#define MLP_T0() long long t_acc[6] = {0, 0, 0, 0, 0, 0}; long long t_last = clock64()
#define MLP_TEND() do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_mlp_t[k_], (unsigned long long)t_acc[k_]); } while (0)
#else
#define MLP_T(k) do { } while (0)
#define MLP_T0() do { } while (0)
#define MLP_TEND() do { } while (0)
#endif

template <int N_IN, int N_HIDDEN>
struct BwdLds {
    using L = LdsW<N_IN, N_HIDDEN>;
    static constexpr int BWD_WAVES = bwd_waves<N_IN, N_HIDDEN>();
    static constexpr int NXB = (N_IN + 31) / 32;                                   // input blocks
    static constexpr bool STAGED = bwd_staged<N_IN, N_HIDDEN>();                   // 7 blocks per wave, the first layer's operands reuse the upper layers'
    static constexpr int NB = STAGED ? 7 : NXB + 4 + (N_HIDDEN == 2 ? 4 : 0) + 1;  // blocks per wave
    static constexpr int W_HALVES = L::SIZE + N_IN * (HID + PAD) + (N_HIDDEN == 2 ? HID * (HID + PAD) : 0) + HID * (16 + PAD);
    static constexpr int OFF_TR = (W_HALVES + 127) / 128 * 128;
    static constexpr int IMG_BYTES = BWD_WAVES * NB * BLK_BYTES;
    static constexpr int PART_BYTES = BWD_WAVES * (STAGED ? 32 : HID) * (N_IN > HID ? N_IN : HID) * 4;   // one f32 slab per wave of the largest layer (STAGED: of its 32-row halves)
    static constexpr int BYTES = OFF_TR * 2 + (IMG_BYTES > PART_BYTES ? IMG_BYTES : PART_BYTES);
};

template <int N_IN, int N_HIDDEN, int IN_MODE, int OUT_MODE>
__global__ void __launch_bounds__((64 * bwd_waves<N_IN, N_HIDDEN>()))
mlp_bwd_kernel(MlpBwdIO io, const h1* __restrict__ weights, int n_samples) {
    using L = LdsW<N_IN, N_HIDDEN>;
    constexpr int BWD_WAVES = bwd_waves<N_IN, N_HIDDEN>();
    // LDS carve-up (halves unless noted):
    //   forward weights (L::SIZE) | transposed weights W0^T (N_IN, 64) W1^T (64,64) Wo^T (64,16)
    //   | per-wave weight-gradient operand images (B::NB blocks of 2 KB) -- reused for the dW reduction slabs at the end
    constexpr int LDT0 = HID + PAD;                 // W0^T rows = in units, cols = 64 hidden
    constexpr int OFF_T0 = L::SIZE;
    constexpr int OFF_T1 = OFF_T0 + N_IN * LDT0;    // W1^T (64,64)
    constexpr int OFF_TO = OFF_T1 + (N_HIDDEN == 2 ? HID * (HID + PAD) : 0);   // Wo^T (64,16)
    constexpr int LDTO = 16 + PAD;
    using B = BwdLds<N_IN, N_HIDDEN>;
    constexpr int OFF_TR = B::OFF_TR;               // the waves' operand images (256-byte aligned); the dW reduction slabs reuse them
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h1* lds = reinterpret_cast<h1*>(smem_raw);
    float* part = reinterpret_cast<float*>(lds + OFF_TR);

    MLP_T0();
    const int lane = threadIdx.x & 63, i = lane & 31, hh = lane >> 5;
    const int wave = threadIdx.x >> 6;
    char* img = reinterpret_cast<char*>(lds + OFF_TR) + wave * B::NB * BLK_BYTES;
    char* img_x = img;                                        // network input            (NXB blocks)
    char* img_dh0 = img_x + B::NXB * BLK_BYTES;               // dL/d hidden 0 (pre-ReLU)   (2)
    char* img_h0 = img_dh0 + 2 * BLK_BYTES;                   // hidden 0                   (2)
    char* img_dh1 = img_h0 + 2 * BLK_BYTES;                   // N_HIDDEN == 2 only         (2)
    char* img_h1 = img_dh1 + 2 * BLK_BYTES;                   //                            (2)
    char* img_dy = img_h0 + (N_HIDDEN == 2 ? 6 : 2) * BLK_BYTES;   // dL/d output (16 units; units 16..31 feed dW rows nobody stores)
    const TrAddr ta = tr_addr(lane);
    const int n_eff = io.active ? min(*io.n_active, n_samples) : n_samples;
    const int n_tiles = (n_eff + TILE - 1) / TILE;
    const float loss_scale = io.loss_scale * (io.loss_scale_dev != nullptr ? *io.loss_scale_dev : 1.0f);
    float din_max = 0.f;                                      // largest |input gradient| this lane converts to f16 (overflow guard)

    f32x16 gW0[2][N_IN / 32 > 0 ? N_IN / 32 : 1];
    f32x16 gW1[2][2];
    f32x16 gWo[1][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < (N_IN / 32 > 0 ? N_IN / 32 : 1); ++n) gW0[m][n] = zero16();
        gW1[m][0] = zero16(); gW1[m][1] = zero16();
    }
    gWo[0][0] = zero16(); gWo[0][1] = zero16();

    // two-deep software pipeline over the wave's tiles: sample id of tile t+2 and raw inputs of tile t+1 are in flight while
    // tile t is computed
    const int tile_stride = gridDim.x * BWD_WAVES;
    // Sample id of tile t for this lane, loaded unconditionally from a clamped position (without a list: a dummy word of the
    // weight blob, ignored).  Lanes past the end get the last sample; a garbage id (n_eff == 0) is clamped into the arrays.
    const bool has_list = io.active != nullptr;                      // wave-uniform
    const long long j_last = n_eff > 0 ? n_eff - 1 : 0;
    const int32_t* idx_src = has_list ? io.active : reinterpret_cast<const int32_t*>(weights);
    auto raw_index = [&](int t) -> int {
        const long long jj = (long long)t * TILE + i;
        return idx_src[has_list ? (jj < j_last ? jj : j_last) : 0];
    };
    auto fetch = [&](int t, int raw, BwdRaw<N_IN>& r) {
        const long long jj = (long long)t * TILE + i;                // compact position
        const bool vv = t < n_tiles && jj < n_eff;
        const long long jc = jj < j_last ? jj : j_last;
        long long sc = has_list ? (long long)raw : jc;
        sc = sc < 0 ? 0 : (sc < n_samples ? sc : (long long)n_samples - 1);
        bwd_fetch<N_IN, IN_MODE, OUT_MODE>(io, jc, sc, vv, n_samples, hh, r);
    };
    BwdRaw<N_IN> cur, nxt;
    const int t0 = blockIdx.x * BWD_WAVES + wave;
    const int raw0 = raw_index(t0);                    // in flight while the weights are staged
    int raw_pre = raw_index(t0 + tile_stride);
    if (NGP_MLP_STAGE_ONCE) {
        stage_bwd_weights<N_IN, N_HIDDEN, 64 * BWD_WAVES>(weights, lds, OFF_T0, OFF_T1, OFF_TO);
    } else {
        stage_fwd_weights<N_IN, N_HIDDEN>(weights, lds);
        stage_weights(weights, lds + OFF_T0, HID, N_IN, true);
        if (N_HIDDEN == 2) stage_weights(weights + L::G_W1, lds + OFF_T1, HID, HID, true);
        stage_weights(weights + L::G_WO, lds + OFF_TO, 16, HID, true);
    }
    fetch(t0, raw0, cur);
    __syncthreads();
    MLP_T(0);                                          // prologue: weights staged, first tile fetched
    for (int tile = t0; tile < n_tiles; tile += tile_stride) {
        fetch(tile + tile_stride, raw_pre, nxt);                     // raw inputs of the next tile ...
        raw_pre = raw_index(tile + 2 * tile_stride);                 // ... and the sample ids of the one after it are in flight
        const long long j = min((long long)tile * TILE + i, j_last);   // compact position (lanes past the end: the last sample, gradient forced to zero)
        const bool valid = cur.valid;
        // ---- forward recompute ----
        half8_t xb[N_IN / 16];
        bwd_input<N_IN, IN_MODE>(cur, hh, xb);
        f32x16 acc[2];
        half8_t h0b[4], h1b[4];
        layer_in<N_IN>(lds + L::OFF_W0, xb, i, hh, acc);
        acc_to_frag<true>(acc, h0b);
        if (N_HIDDEN == 2) {
            layer_hid<2>(lds + L::OFF_W1, L::LDH, 64, h0b, i, hh, acc);
            acc_to_frag<true>(acc, h1b);
        }
        const half8_t (&hlast)[4] = (N_HIDDEN == 2) ? h1b : h0b;
        // ---- output gradient (D order, rows = 16 output units, registers 0..7) ----
        // The pre-activation output is recomputed where the output non-linearity needs it; the
        // condition is wave-uniform so the MFMAs stay in uniform control flow.
        const bool need_out = (OUT_MODE == OUT_RGB) || (OUT_MODE == OUT_DENSITY && io.dL_dsigmas != nullptr) ||
                              (OUT_MODE == OUT_PLAIN && io.fwd.out_act == 1);
        f32x16 o[1];
        o[0] = zero16();
        if (need_out) layer_hid<1>(lds + L::OFF_WO, L::LDH, 16, hlast, i, hh, o);
        half8_t dyb[1];
        {
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            dyb[0] = z;
            if (valid) {
                if (OUT_MODE == OUT_RGB) {
                    if (hh == 0) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float sg = sigmoidf(o[0][c]);
                            dyb[0][c] = (h1)(cur.seed[c] * loss_scale * sg * (1.0f - sg));
                        }
                    }
                } else {
                    float g[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) g[r] = (float)cur.dout[r];
                    if (OUT_MODE == OUT_DENSITY) {
                        if (hh == 0 && io.dL_dsigmas) {
                            // TruncExp backward (custom_functions.py:168-173) on the f16 h[0]
                            const float h0 = (float)(h1)o[0][0];
                            g[0] += cur.seed[0] * loss_scale * __expf(fminf(fmaxf(h0, -15.f), 15.f));
                        }
                    } else if (io.fwd.out_act == 1) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) { const float sg = sigmoidf(o[0][r]); g[r] *= sg * (1.0f - sg); }
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) dyb[0][r] = (h1)g[r];
                }
            }
        }
        MLP_T(1);                                      // forward recompute + output gradient
        // ---- dgrad: output layer -> last hidden ----
        half8_t dh1b[4], dh0b[4];
        {
            // dH^T (64 x samples) = Wo^T (64 x 16) * dY^T : one K chunk (16 output units, D order)
            f32x16 d[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                d[m] = mfma(ldsA_dl(lds + OFF_TO, LDTO, 32 * m + i, true, 0, hh), dyb[0], zero16());
            half8_t (&dst)[4] = (N_HIDDEN == 2) ? dh1b : dh0b;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    dst[2 * m + c2] = relu_bwd(d_to_b<false>(d[m], c2), hlast[2 * m + c2]);
                }
        }
        if (N_HIDDEN == 2) {
            f32x16 d[2];
            d[0] = zero16(); d[1] = zero16();
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    d[m] = mfma(ldsA_dl(lds + OFF_T1, HID + PAD, 32 * m + i, true, c, hh), dh1b[c], d[m]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    dh0b[2 * m + c2] = relu_bwd(d_to_b<false>(d[m], c2), h0b[2 * m + c2]);
                }
        }
        // ---- dgrad into the network input ----
        if (io.dL_din) {
            constexpr int MT_IN = (N_IN + 31) / 32;
            f32x16 d[MT_IN];
#pragma unroll
            for (int m = 0; m < MT_IN; ++m) d[m] = zero16();
            // IN_SH_H only needs input rows 16..31 (the h half); map them to tile rows 0..15
            const int row_off = (IN_MODE == IN_SH_H) ? 16 : 0;
            const int n_rows = (IN_MODE == IN_SH_H) ? 16 : N_IN;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int m = 0; m < MT_IN; ++m)
                    d[m] = mfma(ldsA_dl(lds + OFF_T0, LDT0, row_off + 32 * m + i, (32 * m + i) < n_rows, c, hh), dh0b[c], d[m]);
            if (valid) {
                if (IN_MODE == IN_LEVELMAJOR) {
                    // row (r&3)+8(r>>2)+4hh of tile 0 = feature index; pairs (r, r+1) are one level
                    half2_t* df = reinterpret_cast<half2_t*>(io.dL_din);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int f = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        half2_t v; v[0] = (h1)d[0][r]; v[1] = (h1)d[0][r + 1];
                        din_max = fmaxf(din_max, fmaxf(fabsf(d[0][r]), fabsf(d[0][r + 1])));
                        __builtin_nontemporal_store(v, df + (size_t)(f >> 1) * n_samples + j);
                    }
                } else if (IN_MODE == IN_SH_H) {
                    half4_t lo, hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = (h1)d[0][e]; hi[e] = (h1)d[0][4 + e]; }
                    *reinterpret_cast<half4_t*>(io.dL_din + j * 16 + 4 * hh) = lo;
                    *reinterpret_cast<half4_t*>(io.dL_din + j * 16 + 8 + 4 * hh) = hi;
                } else {
#pragma unroll
                    for (int m = 0; m < MT_IN; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int u = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
                            if (u < N_IN) io.dL_din[j * N_IN + u] = (h1)d[m][r];
                        }
                }
            }
        }
        MLP_T(2);                                      // dgrad + input-gradient store
        // ---- wgrad (samples on K: through the wave-private LDS image, read back transposed) ----
        if (!B::STAGED) {
            wave_lds_sync();                              // the previous tile's reads are behind us (DS ops of a wave run in order)
            tr_put_nat<N_IN / 16>(img_x, xb, ta);         // N_IN == 16: units 16..31 of the block feed dW columns nobody stores
            tr_put_dl<4>(img_dh0, dh0b, ta);
            tr_put_dl<4>(img_h0, h0b, ta);
            if (N_HIDDEN == 2) {
                tr_put_dl<4>(img_dh1, dh1b, ta);
                tr_put_dl<4>(img_h1, h1b, ta);
            }
            tr_put_dl<1>(img_dy, dyb, ta);
            wave_lds_sync();
            MLP_T(3);                                  // image stores
            wgrad_tile<2, B::NXB>(img_dh0, img_x, ta, gW0);                               // dW0 (64 x N_IN) = dH0^T X
            if (N_HIDDEN == 2) wgrad_tile<2, 2>(img_dh1, img_h0, ta, gW1);                // dW1 (64 x 64)   = dH1^T H0
            wgrad_tile<1, 2>(img_dy, (N_HIDDEN == 2) ? img_h1 : img_h0, ta, gWo);         // dWo (16 x 64)   = dY^T Hlast
        } else {
            // 7 blocks: the two upper layers' operands at once, then the first layer's through the blocks they leave behind
            char* img_b = img;
            wave_lds_sync();
            tr_put_dl<1>(img_b, dyb, ta);                                  // block 0 (units 16..31: dW rows nobody stores)
            tr_put_dl<4>(img_b + 1 * BLK_BYTES, h1b, ta);                  // blocks 1-2
            tr_put_dl<4>(img_b + 3 * BLK_BYTES, dh1b, ta);                 // blocks 3-4
            tr_put_dl<4>(img_b + 5 * BLK_BYTES, h0b, ta);                  // blocks 5-6
            wave_lds_sync();
            MLP_T(3);
            wgrad_tile<1, 2>(img_b, img_b + 1 * BLK_BYTES, ta, gWo);
            wgrad_tile<2, 2>(img_b + 3 * BLK_BYTES, img_b + 5 * BLK_BYTES, ta, gW1);
            wave_lds_sync();
            tr_put_dl<4>(img_b, dh0b, ta);                                 // blocks 0-1
            tr_put_nat<N_IN / 16>(img_b + 2 * BLK_BYTES, xb, ta);          // block 2
            wave_lds_sync();
            wgrad_tile<2, B::NXB>(img_b, img_b + 2 * BLK_BYTES, ta, gW0);
        }
        MLP_T(4);                                      // wgrad
        cur = nxt;
    }
    // ---- reduce the waves' dW layer by layer into the workgroup's partial row ----
    __syncthreads();                                      // the slabs reuse the operand images: every wave is done with them
    constexpr int NT0 = (N_IN / 32 > 0 ? N_IN / 32 : 1);
    float* out = io.wgrad_partial + (size_t)blockIdx.x * L::G_SIZE;
    float nonfinite = 0.f;
    wgrad_reduce_layer<2, NT0, BWD_WAVES, B::STAGED>(part, HID, N_IN, wave, i, hh, gW0, out, nonfinite);
    if (N_HIDDEN == 2) wgrad_reduce_layer<2, 2, BWD_WAVES, B::STAGED>(part, HID, HID, wave, i, hh, gW1, out + L::G_W1, nonfinite);
    wgrad_reduce_layer<1, 2, BWD_WAVES, B::STAGED>(part, 16, HID, wave, i, hh, gWo, out + L::G_WO, nonfinite);
    // (NaN != 0: taken exactly when a sum was inf / NaN) ... or a feature gradient left the f16 range on its way to the table backward
    // (finite in the f32 accumulator, inf as the half the scatter reads: the dW sums, f32, do not see that one)
    if (io.nonfinite != nullptr && (nonfinite != 0.f || din_max > (io.din_limit > 0.f ? io.din_limit : 65504.f))) atomicOr(io.nonfinite, 1);
    if (io.nonfinite_clear != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *io.nonfinite_clear = 0;
    MLP_T(5);                                          // epilogue
    MLP_TEND();
}

// ------------------------------------------------------------------------------------------
// fused field backward (NGP_FIELD_BWD_FUSED): the colour net's and the density net's backward chained in ONE launch.  What the two
// mlp_bwd_kernel launches hand over through memory -- dL/dh (S,16) f16 out and in, h (S,16) in -- stays in registers: the
// density net's forward is recomputed from the features (bit for bit the h the forward stored), the colour net's input gradient
// is rounded to f16 exactly as the (S,16) store rounded it and becomes the density net's output gradient in the D order it is
// produced in.  One prologue (both nets' weights + transposes staged once), one epilogue, one pass over the active list.
// The twelve dW accumulator tiles of the two nets (192 registers) are pinned in AGPRs: the wgrad MFMAs are issued as inline
// assembly with "+a" operands, which leaves the 256 VGPRs to forward / dgrad (-amdgpu-mfma-vgpr-form keeps THEIR results in VGPRs).
// Results: dfeats bit-identical to the two-kernel path; dW partial rows differ in summation order (4 waves of tiles instead of 8
// for the density net).
// ------------------------------------------------------------------------------------------
#ifndef NGP_FIELD_BWD_FUSED
#define NGP_FIELD_BWD_FUSED 1          // 0: ngp_field_bwd = the two mlp_bwd_kernel launches (A/B builds; tests reach them through ngp_field_bwd_two_launches)
#endif
struct FieldBwdIO {
    const h1* feats;          // [16][S] half2
    const float* dirs;        // (S,3)
    const float* dL_dsigmas;  // (S) f32 unscaled
    const float* dL_drgbs;    // (S,3) f32 unscaled
    float loss_scale;
    const float* loss_scale_dev;   // optional device-side factor on loss_scale (the dynamic loss scale), read once per launch
    float din_limit;               // a feature gradient beyond this magnitude raises `nonfinite` (0: 65504, the f16 range; a data-parallel rank: 65504 / world)
    h1* dfeats;               // [16][S] half2, by compact position
    float* wgrad_density;     // (gridDim.x, 3072)
    float* wgrad_rgb;         // (gridDim.x, 7168)
    const int32_t* active;    // optional compaction, as MlpBwdIO
    const int32_t* n_active;
    int32_t* nonfinite;
    int32_t* nonfinite_clear;
};
struct FieldBwdRaw {
    bool valid;
    half8_t in[2];
    float dir[3];
    float seed_rgb[3];
    float seed_sig;
};
struct FieldBwdLds {
    using LD = LdsW<32, 1>;
    using LR = LdsW<32, 2>;
    static constexpr int LDT = HID + PAD, LDTO = 16 + PAD;
    // density: forward image | W0^T (32, 64) | Wo^T (64, 16);  colour: forward image | W0^T (32, 64) | W1^T (64, 64) | Wo^T (64, 16)
    static constexpr int D_T0 = LD::SIZE, D_TO = D_T0 + 32 * LDT, D_SIZE = D_TO + HID * LDTO;
    static constexpr int R_T0 = LR::SIZE, R_T1 = R_T0 + 32 * LDT, R_TO = R_T1 + HID * LDT, R_SIZE = R_TO + HID * LDTO;
    static constexpr int OFF_R = (D_SIZE + 7) / 8 * 8;
    static constexpr int OFF_TR = (OFF_R + R_SIZE + 127) / 128 * 128;
    static constexpr int FWAVES = 4, NB = 10;
    static constexpr int IMG_BYTES = FWAVES * NB * BLK_BYTES;                  // 80 KB; the dW reduction slabs (4 x 16 KB) reuse it
    static constexpr int BYTES = OFF_TR * 2 + IMG_BYTES;
};

// dW tile accumulate with the accumulator in AGPRs
__device__ __forceinline__ void mfma_agpr(f32x16& acc, half8_t a, half8_t b) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int MT, int NT>
__device__ __forceinline__ void wgrad_tile_agpr(const char* dy, const char* x, const TrAddr& ta, f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        half8_t a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = tr_get(dy + m * BLK_BYTES, c, ta);
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = tr_get(x + n * BLK_BYTES, c, ta);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) mfma_agpr(acc[m][n], a[m], b[n]);
    }
}
// D-order 16-unit fragment (register e of lane-half hh = unit 4hh + (e&3) + 8(e>>2)) -> natural order (unit 8hh + e): the lane
// halves exchange two registers each (v_permlane32_swap: upper half of the first operand <-> lower half of the second)
__device__ __forceinline__ half8_t d_order_to_natural(half8_t v) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t w = __builtin_bit_cast(u32x4_t, v);
    const auto p02 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
    const auto p13 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
    w[0] = p02[0]; w[2] = p02[1]; w[1] = p13[0]; w[3] = p13[1];
    return __builtin_bit_cast(half8_t, w);
}

__global__ void __launch_bounds__(64 * FieldBwdLds::FWAVES)
field_bwd_kernel(FieldBwdIO io, const h1* __restrict__ density_w, const h1* __restrict__ rgb_w, int n_samples) {
    using F = FieldBwdLds;
    using LD = F::LD;
    using LR = F::LR;
    constexpr int FW = F::FWAVES;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h1* ldsd = reinterpret_cast<h1*>(smem_raw);
    h1* ldsr = ldsd + F::OFF_R;
    float* part = reinterpret_cast<float*>(ldsd + F::OFF_TR);

    MLP_T0();
    const int lane = threadIdx.x & 63, i = lane & 31, hh = lane >> 5;
    const int wave = threadIdx.x >> 6;
    char* img = reinterpret_cast<char*>(ldsd + F::OFF_TR) + wave * F::NB * BLK_BYTES;
    // colour net: x (1 block) dh0 (2) h0 (2) dh1 (2) h1 (2) dy (1); the density net's operands then reuse the first six
    char* img_x = img;
    char* img_dh0 = img + 1 * BLK_BYTES;
    char* img_h0 = img + 3 * BLK_BYTES;
    char* img_dh1 = img + 5 * BLK_BYTES;
    char* img_h1 = img + 7 * BLK_BYTES;
    char* img_dy = img + 9 * BLK_BYTES;
    char* imd_x = img;
    char* imd_dh0 = img + 1 * BLK_BYTES;
    char* imd_h0 = img + 3 * BLK_BYTES;
    char* imd_dy = img + 5 * BLK_BYTES;
    const TrAddr ta = tr_addr(lane);
    const int n_eff = io.active ? min(*io.n_active, n_samples) : n_samples;
    const int n_tiles = (n_eff + TILE - 1) / TILE;
    const float loss_scale = io.loss_scale * (io.loss_scale_dev != nullptr ? *io.loss_scale_dev : 1.0f);
    float din_max = 0.f;                                      // largest |feature gradient| this lane converts to f16 (overflow guard)

    f32x16 gR0[2][1], gR1[2][2], gRo[1][2], gD0[2][1], gDo[1][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        gR0[m][0] = zero16(); gD0[m][0] = zero16();
        gR1[m][0] = zero16(); gR1[m][1] = zero16();
    }
    gRo[0][0] = zero16(); gRo[0][1] = zero16(); gDo[0][0] = zero16(); gDo[0][1] = zero16();

    const int tile_stride = gridDim.x * FW;
    const bool has_list = io.active != nullptr;
    const long long j_last = n_eff > 0 ? n_eff - 1 : 0;
    const int32_t* idx_src = has_list ? io.active : reinterpret_cast<const int32_t*>(density_w);
    auto raw_index = [&](int t) -> int {
        const long long jj = (long long)t * TILE + i;
        return idx_src[has_list ? (jj < j_last ? jj : j_last) : 0];
    };
    const half2_t* fp = reinterpret_cast<const half2_t*>(io.feats);
    auto fetch = [&](int t, int raw, FieldBwdRaw& r) {
        const long long jj = (long long)t * TILE + i;
        r.valid = t < n_tiles && jj < n_eff;
        const long long jc = jj < j_last ? jj : j_last;
        long long sc = has_list ? (long long)raw : jc;
        sc = sc < 0 ? 0 : (sc < n_samples ? sc : (long long)n_samples - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half2_t v = __builtin_nontemporal_load(fp + (size_t)(8 * c + 4 * hh + q) * n_samples + sc);
                r.in[c][2 * q] = v[0]; r.in[c][2 * q + 1] = v[1];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) { r.dir[c] = io.dirs[3 * sc + c]; r.seed_rgb[c] = io.dL_drgbs[3 * sc + c]; }
        r.seed_sig = io.dL_dsigmas[sc];
    };
    FieldBwdRaw cur, nxt;
    const int t0 = blockIdx.x * FW + wave;
    const int raw0 = raw_index(t0);
    int raw_pre = raw_index(t0 + tile_stride);
    stage_bwd_weights<32, 1, 64 * FW>(density_w, ldsd, F::D_T0, 0, F::D_TO);
    stage_bwd_weights<32, 2, 64 * FW>(rgb_w, ldsr, F::R_T0, F::R_T1, F::R_TO);
    fetch(t0, raw0, cur);
    __syncthreads();
    MLP_T(0);
    for (int tile = t0; tile < n_tiles; tile += tile_stride) {
        fetch(tile + tile_stride, raw_pre, nxt);
        raw_pre = raw_index(tile + 2 * tile_stride);
        const long long j = min((long long)tile * TILE + i, j_last);
        const bool valid = cur.valid;
        // ---- density net forward: h (f16, D order) ----
        half8_t xd[2] = {cur.in[0], cur.in[1]};
        f32x16 acc[2];
        half8_t h0d[4];
        layer_in<32>(ldsd + LD::OFF_W0, xd, i, hh, acc);
        acc_to_frag<true>(acc, h0d);
        f32x16 od[1];
        layer_hid<1>(ldsd + LD::OFF_WO, LD::LDH, 16, h0d, i, hh, od);
        half8_t hfrag;
#pragma unroll
        for (int e = 0; e < 8; ++e) hfrag[e] = (h1)od[0][e];
        // ---- colour net forward, operands as the stand-alone kernel forms them (h from its (S,16) row: natural order) ----
        half8_t xr[2];
        xr[0] = sh4_fragment(cur.dir[0], cur.dir[1], cur.dir[2], hh);
        xr[1] = d_order_to_natural(hfrag);
        half8_t h0r[4], h1r[4];
        layer_in<32>(ldsr + LR::OFF_W0, xr, i, hh, acc);
        acc_to_frag<true>(acc, h0r);
        layer_hid<2>(ldsr + LR::OFF_W1, LR::LDH, 64, h0r, i, hh, acc);
        acc_to_frag<true>(acc, h1r);
        f32x16 oc[1];
        layer_hid<1>(ldsr + LR::OFF_WO, LR::LDH, 16, h1r, i, hh, oc);
        half8_t dyr[1];
        {
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            dyr[0] = z;
            if (valid && hh == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float sg = sigmoidf(oc[0][c]);
                    dyr[0][c] = (h1)(cur.seed_rgb[c] * loss_scale * sg * (1.0f - sg));
                }
            }
        }
        MLP_T(1);
        // ---- colour net dgrad ----
        half8_t dh1r[4], dh0r[4];
        {
            f32x16 d[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                d[m] = mfma(ldsA_dl(ldsr + F::R_TO, F::LDTO, 32 * m + i, true, 0, hh), dyr[0], zero16());
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) dh1r[2 * m + c2] = relu_bwd(d_to_b<false>(d[m], c2), h1r[2 * m + c2]);
            d[0] = zero16(); d[1] = zero16();
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    d[m] = mfma(ldsA_dl(ldsr + F::R_T1, F::LDT, 32 * m + i, true, c, hh), dh1r[c], d[m]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) dh0r[2 * m + c2] = relu_bwd(d_to_b<false>(d[m], c2), h0r[2 * m + c2]);
        }
        // input rows 16..31 of the colour net = h: dL/dh, rounded to f16 as the (S,16) hand-over rounded it (D order)
        f32x16 dhf = zero16();
#pragma unroll
        for (int c = 0; c < 4; ++c)
            dhf = mfma(ldsA_dl(ldsr + F::R_T0, F::LDT, 16 + i, i < 16, c, hh), dh0r[c], dhf);
        MLP_T(2);
        // ---- colour net wgrad ----
        wave_lds_sync();
        tr_put_nat<2>(img_x, xr, ta);
        tr_put_dl<4>(img_dh0, dh0r, ta);
        tr_put_dl<4>(img_h0, h0r, ta);
        tr_put_dl<4>(img_dh1, dh1r, ta);
        tr_put_dl<4>(img_h1, h1r, ta);
        tr_put_dl<1>(img_dy, dyr, ta);
        wave_lds_sync();
        MLP_T(3);
        wgrad_tile_agpr<2, 1>(img_dh0, img_x, ta, gR0);
        wgrad_tile_agpr<2, 2>(img_dh1, img_h0, ta, gR1);
        wgrad_tile_agpr<1, 2>(img_dy, img_h1, ta, gRo);
        MLP_T(4);
        // ---- density net: output gradient = dL/dh + TruncExp backward of the sigma seed (custom_functions.py:168-173) ----
        half8_t dyd[1];
        {
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            dyd[0] = z;
            if (valid) {
                float g[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) g[r] = (float)(h1)dhf[r];
                if (hh == 0) g[0] += cur.seed_sig * loss_scale * __expf(fminf(fmaxf((float)hfrag[0], -15.f), 15.f));
#pragma unroll
                for (int r = 0; r < 8; ++r) dyd[0][r] = (h1)g[r];
            }
        }
        half8_t dh0d[4];
        {
            f32x16 d[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                d[m] = mfma(ldsA_dl(ldsd + F::D_TO, F::LDTO, 32 * m + i, true, 0, hh), dyd[0], zero16());
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) dh0d[2 * m + c2] = relu_bwd(d_to_b<false>(d[m], c2), h0d[2 * m + c2]);
        }
        {
            f32x16 d = zero16();
#pragma unroll
            for (int c = 0; c < 4; ++c)
                d = mfma(ldsA_dl(ldsd + F::D_T0, F::LDT, i, true, c, hh), dh0d[c], d);
            if (valid) {
                half2_t* df = reinterpret_cast<half2_t*>(io.dfeats);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int f = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    half2_t v; v[0] = (h1)d[r]; v[1] = (h1)d[r + 1];
                    din_max = fmaxf(din_max, fmaxf(fabsf(d[r]), fabsf(d[r + 1])));
                    __builtin_nontemporal_store(v, df + (size_t)(f >> 1) * n_samples + j);
                }
            }
        }
        MLP_T(2);
        // ---- density net wgrad through the blocks the colour net's operands have been read from ----
        wave_lds_sync();
        tr_put_nat<2>(imd_x, xd, ta);
        tr_put_dl<4>(imd_dh0, dh0d, ta);
        tr_put_dl<4>(imd_h0, h0d, ta);
        tr_put_dl<1>(imd_dy, dyd, ta);
        wave_lds_sync();
        MLP_T(3);
        wgrad_tile_agpr<2, 1>(imd_dh0, imd_x, ta, gD0);
        wgrad_tile_agpr<1, 2>(imd_dy, imd_h0, ta, gDo);
        MLP_T(4);
        cur = nxt;
    }
    // ---- reduce the waves' dW layer by layer into the workgroup's partial rows ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last inline-assembly MFMAs retire before their AGPRs are read
    __syncthreads();
    float* outd = io.wgrad_density + (size_t)blockIdx.x * LD::G_SIZE;
    float* outr = io.wgrad_rgb + (size_t)blockIdx.x * LR::G_SIZE;
    float nonfinite = 0.f;
    wgrad_reduce_layer<2, 1, FW, false>(part, HID, 32, wave, i, hh, gR0, outr, nonfinite);
    wgrad_reduce_layer<2, 2, FW, false>(part, HID, HID, wave, i, hh, gR1, outr + LR::G_W1, nonfinite);
    wgrad_reduce_layer<1, 2, FW, false>(part, 16, HID, wave, i, hh, gRo, outr + LR::G_WO, nonfinite);
    wgrad_reduce_layer<2, 1, FW, false>(part, HID, 32, wave, i, hh, gD0, outd, nonfinite);
    wgrad_reduce_layer<1, 2, FW, false>(part, 16, HID, wave, i, hh, gDo, outd + LD::G_WO, nonfinite);
    if (io.nonfinite != nullptr && (nonfinite != 0.f || din_max > (io.din_limit > 0.f ? io.din_limit : 65504.f))) atomicOr(io.nonfinite, 1);      // (as mlp_bwd_kernel)
    if (io.nonfinite_clear != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *io.nonfinite_clear = 0;
    MLP_T(5);
    MLP_TEND();
}

template <int N_IN, int N_HIDDEN>
constexpr int fwd_smem_bytes() { return LdsW<N_IN, N_HIDDEN>::SIZE * 2; }
template <int N_IN, int N_HIDDEN>
constexpr int bwd_smem_bytes() { return BwdLds<N_IN, N_HIDDEN>::BYTES; }

int fwd_grid(int n_samples) {
    const int n_tiles = (n_samples + TILE - 1) / TILE;
    const int blocks = (n_tiles + WAVES - 1) / WAVES;
    // workgroups per CU: the per-workgroup weight staging (24 KB) wants several tiles per wave, the latency of a tile's loads
    // wants several waves per SIMD (2 workgroups = 2 waves per SIMD measured 75 us per 1.3 M samples in the frame loop)
    constexpr int cap = 512;
    return blocks < cap ? (blocks < 1 ? 1 : blocks) : cap;
}
int bwd_grid(int n_samples) {
    const int n_tiles = (n_samples + TILE - 1) / TILE;
    const int blocks = (n_tiles + 3) / 4;                    // (the same count for every network: the callers' partial rows)
    return blocks < 256 ? (blocks < 1 ? 1 : blocks) : 256;   // one workgroup per CU; bounds the partial buffer
}

template <int N_IN, int N_HIDDEN, int IN_MODE, int OUT_MODE>
int launch_fwd(const MlpIO& io, const h1* w, int n_samples, hipStream_t st) {
    constexpr int smem = fwd_smem_bytes<N_IN, N_HIDDEN>();
    mlp_fwd_kernel<N_IN, N_HIDDEN, IN_MODE, OUT_MODE><<<dim3(fwd_grid(n_samples)), dim3(64 * WAVES), smem, st>>>(io, w, n_samples);
    return NGP_LAUNCH_RESULT();
}
template <int N_IN, int N_HIDDEN, int IN_MODE, int OUT_MODE>
int launch_bwd(const MlpBwdIO& io, const h1* w, int n_samples, hipStream_t st) {
    constexpr int smem = bwd_smem_bytes<N_IN, N_HIDDEN>();
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (reinterpret_cast<uintptr_t>(io.wgrad_partial) & 15) return NGP_EINVAL;     // partial rows are stored 16 bytes at a time
    auto kern = mlp_bwd_kernel<N_IN, N_HIDDEN, IN_MODE, OUT_MODE>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
    }
    kern<<<dim3(bwd_grid(n_samples)), dim3(64 * bwd_waves<N_IN, N_HIDDEN>()), smem, st>>>(io, w, n_samples);
    return NGP_LAUNCH_RESULT();
}

__global__ void __launch_bounds__(256)
sh4_kernel(const float* __restrict__ d01, int n, h1* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    // tcnn.Encoding receives (d+1)/2 and maps it back (networks.py:144; spherical_harmonics.h)
    const float x = d01[3 * s] * 2.f - 1.f, y = d01[3 * s + 1] * 2.f - 1.f, z = d01[3 * s + 2] * 2.f - 1.f;
    float sh[16];
    sh4(x, y, z, sh);
    half8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (h1)sh[e]; b[e] = (h1)sh[8 + e]; }
    *reinterpret_cast<half8_t*>(out + (size_t)s * 16) = a;
    *reinterpret_cast<half8_t*>(out + (size_t)s * 16 + 8) = b;
}

// d SH / d (x,y,z) contracted with dL/dSH (tiny-cuda-nn SphericalHarmonics backward-input);
// the encoding receives (d+1)/2, hence the factor 2 (networks.py:144).
__global__ void __launch_bounds__(256)
sh4_bwd_kernel(const float* __restrict__ d01, const h1* __restrict__ dL_dsh, int n, float out_scale, float* __restrict__ dL_dd01) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float x = d01[3 * s] * 2.f - 1.f, y = d01[3 * s + 1] * 2.f - 1.f, z = d01[3 * s + 2] * 2.f - 1.f;
    float g[16];
    const half8_t a = *reinterpret_cast<const half8_t*>(dL_dsh + (size_t)s * 16);
    const half8_t b = *reinterpret_cast<const half8_t*>(dL_dsh + (size_t)s * 16 + 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[e] = (float)a[e]; g[8 + e] = (float)b[e]; }
    const float A = 0.48860251190291987f, B = 1.0925484305920792f, Cc = 0.94617469575755997f, E = 0.54627421529603959f,
                F = 0.59004358992664352f, G = 2.8906114426405538f, H = 0.45704579946446572f, K = 0.3731763325901154f,
                M = 1.4453057213202769f;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    gy += g[1] * -A;
    gz += g[2] * A;
    gx += g[3] * -A;
    gx += g[4] * B * y; gy += g[4] * B * x;
    gy += g[5] * -B * z; gz += g[5] * -B * y;
    gz += g[6] * 2.f * Cc * z;
    gx += g[7] * -B * z; gz += g[7] * -B * x;
    gx += g[8] * 2.f * E * x; gy += g[8] * -2.f * E * y;
    gx += g[9] * -6.f * F * x * y; gy += g[9] * F * (-3.f * x2 + 3.f * y2);
    gx += g[10] * G * y * z; gy += g[10] * G * x * z; gz += g[10] * G * x * y;
    gy += g[11] * H * (1.f - 5.f * z2); gz += g[11] * -10.f * H * y * z;
    gz += g[12] * K * (15.f * z2 - 3.f);
    gx += g[13] * H * (1.f - 5.f * z2); gz += g[13] * -10.f * H * x * z;
    gx += g[14] * 2.f * M * x * z; gy += g[14] * -2.f * M * y * z; gz += g[14] * M * (x2 - y2);
    gx += g[15] * F * (-3.f * x2 + 3.f * y2); gy += g[15] * 6.f * F * x * y;
    dL_dd01[3 * s] = 2.f * gx * out_scale; dL_dd01[3 * s + 1] = 2.f * gy * out_scale; dL_dd01[3 * s + 2] = 2.f * gz * out_scale;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int ngp_density_fwd(const ngp_half* feats, const ngp_half* density_w, int n_samples, float* sigmas,
                    ngp_half* h_out, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(sigmas);
    MlpIO d = {};
    d.in = (const h1*)feats; d.out16 = (h1*)h_out; d.out_ld = 16; d.n_out = 16; d.sigmas = sigmas;
    return launch_fwd<32, 1, IN_LEVELMAJOR, OUT_DENSITY>(d, (const h1*)density_w, n_samples, ngp_stream(stream));
}

int ngp_density_fwd_scatter(const ngp_half* feats, const ngp_half* density_w, int n_samples,
                            const int32_t* scatter_idx, float* sigmas_out, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(scatter_idx); NGP_CHECK_PTR(sigmas_out);
    MlpIO d = {};
    d.in = (const h1*)feats; d.out_ld = 16; d.n_out = 16; d.sigmas = sigmas_out; d.scatter = scatter_idx;
    return launch_fwd<32, 1, IN_LEVELMAJOR, OUT_DENSITY>(d, (const h1*)density_w, n_samples, ngp_stream(stream));
}

int ngp_field_fwd(const ngp_half* feats, const float* dirs, const ngp_half* density_w, const ngp_half* rgb_w,
                  int n_samples, float* sigmas, float* rgbs, ngp_half* h_out, ngp_stream_t stream) {
    return ngp_field_fwd_n(feats, dirs, density_w, rgb_w, n_samples, nullptr, sigmas, rgbs, h_out, stream);
}

int ngp_field_fwd_n(const ngp_half* feats, const float* dirs, const ngp_half* density_w, const ngp_half* rgb_w,
                    int n_samples, const int32_t* n_dev, float* sigmas, float* rgbs, ngp_half* h_out,
                    ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(dirs); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(rgb_w);
    NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs);          // h_out may be NULL (inference: h stays in registers)
    FieldIO io = {};
    io.feats = (const h1*)feats; io.dirs = dirs; io.sigmas = sigmas; io.rgbs = rgbs; io.h_out = (h1*)h_out; io.n_dev = n_dev;
    constexpr int smem = fwd_smem_bytes<32, 1>() + fwd_smem_bytes<32, 2>();
    field_fwd_kernel<false><<<dim3(fwd_grid(n_samples)), dim3(64 * WAVES), smem, ngp_stream(stream)>>>(
        io, (const h1*)density_w, (const h1*)rgb_w, n_samples);
    return NGP_LAUNCH_RESULT();
}

int ngp_field_fwd_list(const ngp_half* feats, const float* dirs, const ngp_half* density_w, const ngp_half* rgb_w,
                       int n_samples, const int32_t* list, int n_list_max, const int32_t* n_list_dev,
                       float* sigmas, float* rgbs, ngp_half* h_out, ngp_stream_t stream) {
    if (n_samples < 0 || n_list_max < 0) return NGP_EINVAL;
    if (n_samples == 0 || n_list_max == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(dirs); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(rgb_w);
    NGP_CHECK_PTR(sigmas); NGP_CHECK_PTR(rgbs); NGP_CHECK_PTR(list);
    FieldIO io = {};
    io.feats = (const h1*)feats; io.dirs = dirs; io.sigmas = sigmas; io.rgbs = rgbs; io.h_out = (h1*)h_out; io.n_dev = n_list_dev;
    io.list = list; io.stride = n_samples;
    constexpr int smem = fwd_smem_bytes<32, 1>() + fwd_smem_bytes<32, 2>();
    field_fwd_kernel<true><<<dim3(fwd_grid(n_list_max)), dim3(64 * WAVES), smem, ngp_stream(stream)>>>(
        io, (const h1*)density_w, (const h1*)rgb_w, n_list_max);
    return NGP_LAUNCH_RESULT();
}

#ifdef NGP_MLP_TIMING
int ngp_debug_mlp_timing(unsigned long long* host_out, int reset) {
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mlp_t), sizeof(g_mlp_t)) != hipSuccess) return NGP_EINVAL;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_t), z, sizeof(z)) != hipSuccess) return NGP_EINVAL; }
    return 0;
}
#endif

int ngp_field_bwd_partials(int n_samples) { return n_samples <= 0 ? 0 : bwd_grid(n_samples); }

// (the colour net's half of ngp_field_bwd: internal)
static int ngp_rgb_bwd(const ngp_half* h, const float* dirs, const ngp_half* rgb_w, const float* dL_drgbs, float loss_scale, const float* loss_scale_dev,
                int n_samples, const int32_t* active_idx, const int32_t* n_active, ngp_half* dL_dh, float* wgrad_partial,
                int32_t* nonfinite, int32_t* nonfinite_clear, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(h); NGP_CHECK_PTR(dirs); NGP_CHECK_PTR(rgb_w); NGP_CHECK_PTR(dL_drgbs); NGP_CHECK_PTR(dL_dh); NGP_CHECK_PTR(wgrad_partial);
    MlpBwdIO r = {};
    r.fwd.in = (const h1*)h; r.fwd.dirs = dirs; r.fwd.n_out = 3;
    r.dL_drgbs = dL_drgbs; r.loss_scale = loss_scale; r.loss_scale_dev = loss_scale_dev; r.dL_din = (h1*)dL_dh; r.wgrad_partial = wgrad_partial;
    if ((active_idx == nullptr) != (n_active == nullptr)) return NGP_EINVAL;
    r.active = active_idx; r.n_active = n_active;
    r.nonfinite = nonfinite; r.nonfinite_clear = nonfinite_clear;
    return launch_bwd<32, 2, IN_SH_H, OUT_RGB>(r, (const h1*)rgb_w, n_samples, ngp_stream(stream));
}

static int density_bwd_guarded(const ngp_half* feats, const ngp_half* density_w, const ngp_half* dL_dh, const float* dL_dsigmas,
                               float loss_scale, const float* loss_scale_dev, float din_limit, int n_samples, const int32_t* active_idx, const int32_t* n_active,
                               ngp_half* dfeats, float* wgrad_partial, int32_t* nonfinite, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(dfeats); NGP_CHECK_PTR(wgrad_partial);
    MlpBwdIO d = {};
    d.fwd.in = (const h1*)feats; d.fwd.n_out = 16; d.fwd.out_ld = 16;
    d.dL_dout16 = (const h1*)dL_dh; d.dout_ld = 16;
    d.dL_dsigmas = dL_dsigmas; d.loss_scale = loss_scale; d.loss_scale_dev = loss_scale_dev; d.din_limit = din_limit; d.dL_din = (h1*)dfeats; d.wgrad_partial = wgrad_partial;
    if ((active_idx == nullptr) != (n_active == nullptr)) return NGP_EINVAL;
    d.active = active_idx; d.n_active = n_active;
    d.nonfinite = nonfinite;
    return launch_bwd<32, 1, IN_LEVELMAJOR, OUT_DENSITY>(d, (const h1*)density_w, n_samples, ngp_stream(stream));
}

int ngp_density_bwd(const ngp_half* feats, const ngp_half* density_w, const ngp_half* dL_dh, const float* dL_dsigmas,
                    float loss_scale, int n_samples, const int32_t* active_idx, const int32_t* n_active,
                    ngp_half* dfeats, float* wgrad_partial, ngp_stream_t stream) {
    return density_bwd_guarded(feats, density_w, dL_dh, dL_dsigmas, loss_scale, nullptr, 0.f, n_samples, active_idx, n_active, dfeats, wgrad_partial, nullptr, stream);
}

int ngp_field_bwd(const ngp_half* feats, const float* dirs, const ngp_half* h, const ngp_half* density_w,
                  const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale,
                  int n_samples, const int32_t* active_idx, const int32_t* n_active,
                  ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial, ngp_stream_t stream) {
    return ngp_field_bwd_guarded(feats, dirs, h, density_w, rgb_w, dL_dsigmas, dL_drgbs, loss_scale, nullptr, 0.f, n_samples, active_idx, n_active, dh_scratch, dfeats,
                                 wgrad_partial, nullptr, 0, stream);
}

// The two-launch form: colour net (writes dL/dh to dh_scratch) then density net (reads it).  nonfinite2 / parity as below.
static int field_bwd_two_launches(const ngp_half* feats, const float* dirs, const ngp_half* h, const ngp_half* density_w,
                                  const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale, const float* loss_scale_dev, float din_limit,
                                  int n_samples, const int32_t* active_idx, const int32_t* n_active,
                                  ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial, int32_t* nonfinite2, int parity, ngp_stream_t stream) {
    const int n_part = bwd_grid(n_samples);
    int32_t* flag = nonfinite2 ? nonfinite2 + (parity & 1) : nullptr;
    const int rc = ngp_rgb_bwd(h, dirs, rgb_w, dL_drgbs, loss_scale, loss_scale_dev, n_samples, active_idx, n_active, dh_scratch,
                               wgrad_partial + (size_t)n_part * NGP_DENSITY_NET_PARAMS, flag, nonfinite2 ? nonfinite2 + ((parity & 1) ^ 1) : nullptr, stream);
    if (rc) return rc;
    return density_bwd_guarded(feats, density_w, dh_scratch, dL_dsigmas, loss_scale, loss_scale_dev, din_limit, n_samples, active_idx, n_active, dfeats,
                               wgrad_partial, flag, stream);
}

// The one-launch form (field_bwd_kernel): h and dL/dh never touch memory.
static int field_bwd_one_launch(const ngp_half* feats, const float* dirs, const ngp_half* density_w,
                                const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale, const float* loss_scale_dev, float din_limit,
                                int n_samples, const int32_t* active_idx, const int32_t* n_active,
                                ngp_half* dfeats, float* wgrad_partial, int32_t* nonfinite2, int parity, ngp_stream_t stream) {
    NGP_CHECK_PTR(feats); NGP_CHECK_PTR(dirs); NGP_CHECK_PTR(density_w); NGP_CHECK_PTR(rgb_w); NGP_CHECK_PTR(dL_dsigmas);
    NGP_CHECK_PTR(dL_drgbs); NGP_CHECK_PTR(dfeats);
    if ((active_idx == nullptr) != (n_active == nullptr)) return NGP_EINVAL;
    if (reinterpret_cast<uintptr_t>(wgrad_partial) & 15) return NGP_EINVAL;     // partial rows are stored 16 bytes at a time
    const int n_part = bwd_grid(n_samples);
    FieldBwdIO io = {};
    io.feats = (const h1*)feats; io.dirs = dirs; io.dL_dsigmas = dL_dsigmas; io.dL_drgbs = dL_drgbs; io.loss_scale = loss_scale; io.loss_scale_dev = loss_scale_dev; io.din_limit = din_limit;
    io.dfeats = (h1*)dfeats; io.wgrad_density = wgrad_partial; io.wgrad_rgb = wgrad_partial + (size_t)n_part * NGP_DENSITY_NET_PARAMS;
    io.active = active_idx; io.n_active = n_active;
    io.nonfinite = nonfinite2 ? nonfinite2 + (parity & 1) : nullptr;
    io.nonfinite_clear = nonfinite2 ? nonfinite2 + ((parity & 1) ^ 1) : nullptr;
    constexpr int smem = FieldBwdLds::BYTES;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_set[64] = {};              // per device: the attribute belongs to the device's code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(field_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    field_bwd_kernel<<<dim3(n_part), dim3(64 * FieldBwdLds::FWAVES), smem, ngp_stream(stream)>>>(io, (const h1*)density_w, (const h1*)rgb_w, n_samples);
    return NGP_LAUNCH_RESULT();
}

// (csrc/ngp_internal.h) the same with the native stepper's overflow guard: nonfinite2 = two device flags; this call ORs 1 into
// nonfinite2[parity] when a weight-gradient sum of either network is inf / NaN -- or a feature gradient leaves the f16 range -- and
// clears nonfinite2[parity ^ 1] (the next step's).  loss_scale_dev (may be NULL): a device-side factor on loss_scale, read by the
// launch (the stepper's dynamic loss scale: GradScaler's scale on top of tiny-cuda-nn's 128).  din_limit: the magnitude beyond which a
// feature gradient raises the flag (0 = 65504; a data-parallel rank passes 65504 / world: the ranks' gradients are summed in f16).
int ngp_field_bwd_guarded(const ngp_half* feats, const float* dirs, const ngp_half* h, const ngp_half* density_w,
                          const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale, const float* loss_scale_dev, float din_limit,
                          int n_samples, const int32_t* active_idx, const int32_t* n_active,
                          ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial, int32_t* nonfinite2, int parity, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(wgrad_partial);
    if (NGP_FIELD_BWD_FUSED)
        return field_bwd_one_launch(feats, dirs, density_w, rgb_w, dL_dsigmas, dL_drgbs, loss_scale, loss_scale_dev, din_limit, n_samples, active_idx, n_active, dfeats,
                                    wgrad_partial, nonfinite2, parity, stream);
    return field_bwd_two_launches(feats, dirs, h, density_w, rgb_w, dL_dsigmas, dL_drgbs, loss_scale, loss_scale_dev, din_limit, n_samples, active_idx, n_active, dh_scratch,
                                  dfeats, wgrad_partial, nonfinite2, parity, stream);
}

// (csrc/ngp_internal.h) 1 when ngp_field_bwd reads the forward's h_out and writes dh_scratch (the two-launch A/B build), 0 when both
// may be NULL: callers that own the buffers (the native stepper) skip the forward's h store.
int ngp_field_bwd_uses_h(void) { return NGP_FIELD_BWD_FUSED ? 0 : 1; }

// (csrc/ngp_internal.h, test hook) ngp_field_bwd as two launches whatever the build's default: the cross-check of the one-launch kernel
// (dfeats must agree bit for bit; the density net's dW partial rows differ in summation order).
int ngp_field_bwd_two_launches(const ngp_half* feats, const float* dirs, const ngp_half* h, const ngp_half* density_w,
                               const ngp_half* rgb_w, const float* dL_dsigmas, const float* dL_drgbs, float loss_scale,
                               int n_samples, const int32_t* active_idx, const int32_t* n_active,
                               ngp_half* dh_scratch, ngp_half* dfeats, float* wgrad_partial, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(wgrad_partial); NGP_CHECK_PTR(h); NGP_CHECK_PTR(dh_scratch);
    return field_bwd_two_launches(feats, dirs, h, density_w, rgb_w, dL_dsigmas, dL_drgbs, loss_scale, nullptr, 0.f, n_samples, active_idx, n_active, dh_scratch,
                                  dfeats, wgrad_partial, nullptr, 0, stream);
}

int ngp_mlp_fwd(const ngp_half* in, const ngp_half* weights, int n_in, int n_hidden, int n_out, int out_act,
                int n_samples, ngp_half* out, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_out < 1 || n_out > 16 || (out_act != 0 && out_act != 1)) return NGP_EUNSUP;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(in); NGP_CHECK_PTR(weights); NGP_CHECK_PTR(out);
    MlpIO io = {};
    io.in = (const h1*)in; io.out16 = (h1*)out; io.out_ld = n_out; io.n_out = n_out; io.out_act = out_act;
    hipStream_t st = ngp_stream(stream);
    const h1* w = (const h1*)weights;
    if (n_in == 16 && n_hidden == 1) return launch_fwd<16, 1, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 16 && n_hidden == 2) return launch_fwd<16, 2, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 32 && n_hidden == 1) return launch_fwd<32, 1, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 32 && n_hidden == 2) return launch_fwd<32, 2, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 64 && n_hidden == 1) return launch_fwd<64, 1, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 64 && n_hidden == 2) return launch_fwd<64, 2, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    return NGP_EUNSUP;
}

int ngp_mlp_bwd_partials(int n_samples) { return n_samples <= 0 ? 0 : bwd_grid(n_samples); }

int ngp_mlp_bwd(const ngp_half* in, const ngp_half* weights, const ngp_half* dL_dout, int n_in, int n_hidden,
                int n_out, int out_act, int n_samples, ngp_half* dL_din, float* wgrad_partial, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_out < 1 || n_out > 16 || (out_act != 0 && out_act != 1)) return NGP_EUNSUP;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(in); NGP_CHECK_PTR(weights); NGP_CHECK_PTR(dL_dout); NGP_CHECK_PTR(wgrad_partial);
    MlpBwdIO io = {};
    io.fwd.in = (const h1*)in; io.fwd.n_out = n_out; io.fwd.out_act = out_act; io.fwd.out_ld = n_out;
    io.dL_dout16 = (const h1*)dL_dout; io.dout_ld = n_out; io.loss_scale = 1.0f;
    io.dL_din = (h1*)dL_din; io.wgrad_partial = wgrad_partial;
    hipStream_t st = ngp_stream(stream);
    const h1* w = (const h1*)weights;
    if (n_in == 16 && n_hidden == 1) return launch_bwd<16, 1, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 16 && n_hidden == 2) return launch_bwd<16, 2, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 32 && n_hidden == 1) return launch_bwd<32, 1, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 32 && n_hidden == 2) return launch_bwd<32, 2, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 64 && n_hidden == 1) return launch_bwd<64, 1, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    if (n_in == 64 && n_hidden == 2) return launch_bwd<64, 2, IN_ROWMAJOR, OUT_PLAIN>(io, w, n_samples, st);
    return NGP_EUNSUP;
}

int ngp_sh4_fwd(const float* dirs01, int n_samples, ngp_half* out, ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(dirs01); NGP_CHECK_PTR(out);
    hipLaunchKernelGGL(sh4_kernel, dim3(ngp_div_up(n_samples, 256)), dim3(256), 0, ngp_stream(stream), dirs01, n_samples, (h1*)out);
    return NGP_LAUNCH_RESULT();
}

int ngp_sh4_bwd(const float* dirs01, const ngp_half* dL_dsh, int n_samples, float out_scale, float* dL_ddirs01,
                ngp_stream_t stream) {
    if (n_samples < 0) return NGP_EINVAL;
    if (n_samples == 0) return 0;
    NGP_CHECK_PTR(dirs01); NGP_CHECK_PTR(dL_dsh); NGP_CHECK_PTR(dL_ddirs01);
    hipLaunchKernelGGL(sh4_bwd_kernel, dim3(ngp_div_up(n_samples, 256)), dim3(256), 0, ngp_stream(stream), dirs01, (const h1*)dL_dsh,
                       n_samples, out_scale, dL_ddirs01);
    return NGP_LAUNCH_RESULT();
}

}  // extern "C"
