// TEST INFRASTRUCTURE.  Minimal CPU stand-in for the CUDA runtime surface that the reference's
// models/csrc/*.cu use, so that those files can be compiled UNMODIFIED IN PLACE by g++ and run
// on the host as the strongest available pin for oracle/ngp_oracle.c (see oracle/build_ref.sh).
// A kernel "launch" runs the kernel body once per (block, thread) index, sequentially.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

struct uint3_ { unsigned int x, y, z; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local uint3_ threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

#define REF_VEC2(T, N) struct N##2 { T x, y; }; static inline N##2 make_##N##2(T x, T y) { N##2 v; v.x = x; v.y = y; return v; }
#define REF_VEC3(T, N) struct N##3 { T x, y, z; }; static inline N##3 make_##N##3(T x, T y, T z) { N##3 v; v.x = x; v.y = y; v.z = z; return v; }
#define REF_VEC4(T, N) struct N##4 { T x, y, z, w; }; static inline N##4 make_##N##4(T x, T y, T z, T w) { N##4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
REF_VEC2(float, float) REF_VEC3(float, float) REF_VEC4(float, float)
REF_VEC2(int, int) REF_VEC3(int, int) REF_VEC4(int, int)
REF_VEC2(unsigned int, uint) REF_VEC3(unsigned int, uint) REF_VEC4(unsigned int, uint)

// integer min/max and rsqrtf exist as CUDA device builtins
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// fast-math exponential: accurate expf here (the oracle makes the same choice); glibc already
// declares a symbol called __expf, hence the rename
static inline float ref_fast_expf(float x) { return expf(x); }
#define __expf ref_fast_expf

// threads run one after another, so a plain read-modify-write is atomic
static inline int atomicAdd(int* p, int v) { const int old = *p; *p += v; return old; }

template <typename Kernel, typename... Args>
static inline void ref_launch(dim3 grid, dim3 block, Kernel kernel, Args... args) {
    gridDim = grid; blockDim = block;
    for (unsigned int by = 0; by < grid.y; ++by)
        for (unsigned int bx = 0; bx < grid.x; ++bx)
            for (unsigned int ty = 0; ty < block.y; ++ty)
                for (unsigned int tx = 0; tx < block.x; ++tx) {
                    blockIdx.x = bx; blockIdx.y = by; blockIdx.z = 0;
                    threadIdx.x = tx; threadIdx.y = ty; threadIdx.z = 0;
                    kernel(args...);
                }
}
