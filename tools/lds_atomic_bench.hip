// Scratch micro-benchmark: cost of LDS atomics on gfx950 as a function of type and active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#define AS3 __attribute__((address_space(3)))

template <int MODE, int ACTIVE, int GROUP = 1>
__global__ void __launch_bounds__(1024) k(uint32_t seed, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* lds = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    uint32_t r = seed + (threadIdx.x / GROUP) * 2654435761u;      // GROUP consecutive lanes share their addresses (same-address conflicts)
    const bool act = (threadIdx.x & 63) < ACTIVE;
    half2_t hv = {(_Float16)0.001f, (_Float16)0.002f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            r = r * 1664525u + 1013904223u;
            const uint32_t idx = (r >> 8) & 32767;
            if (act) {
                if (MODE == 0) __builtin_amdgcn_ds_atomic_fadd_v2f16((AS3 half2_t*)(lds + idx), hv);
                else if (MODE == 1) atomicAdd((float*)(lds + idx), 0.001f);
                else if (MODE == 2) atomicAdd(lds + idx, 3u);
                else if (MODE == 3) lds[idx] = r;          // plain store for reference
                else if (MODE == 4) atomicAdd((unsigned long long*)(lds + (idx & ~1u)), (unsigned long long)r);   // ds_add_u64
                else if (MODE == 5) { atomicAdd(lds + (idx & ~1u), r); atomicAdd(lds + (idx | 1u), r >> 3); }   // two ds_add_u32
            }
        }
    }
    __syncthreads();
    float s = 0; for (int i = threadIdx.x; i < 32768; i += blockDim.x) s += (float)lds[i];
    if (s == 12345.f) out[0] = s;
}

template <int MODE, int ACTIVE, int GROUP = 1>
void run(const char* name) {
    float* out; hipMalloc(&out, 4);
    auto kern = k<MODE, ACTIVE, GROUP>;
    if (GROUP > 1) printf("[%d lanes per address] ", GROUP);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, 1024, 131072>>>(1, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<256, 1024, 131072>>>(2, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr = 16.0 * iters * 8;                  // per CU
    printf("%-12s active %2d/64: %8.1f us  -> %6.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, ACTIVE, ms * 1e3,
           ms * 1e-3 * 2.4e9 / wave_instr);
}
int main() {
    run<0, 64>("pk_add_f16"); run<0, 16>("pk_add_f16"); run<0, 4>("pk_add_f16"); run<0, 1>("pk_add_f16");
    run<1, 64>("add_f32"); run<1, 4>("add_f32");
    run<2, 64>("add_u32"); run<2, 4>("add_u32");
    run<3, 64>("store_b32"); run<3, 4>("store_b32");
    run<4, 64>("add_u64"); run<4, 16>("add_u64"); run<4, 4>("add_u64"); run<4, 2>("add_u64");
    run<5, 64>("2x add_u32"); run<5, 4>("2x add_u32");
    run<0, 2>("pk_add_f16"); run<2, 16>("add_u32"); run<2, 2>("add_u32");
    run<4, 64, 8>("add_u64"); run<4, 64, 64>("add_u64"); run<0, 64, 8>("pk_add_f16"); run<0, 64, 64>("pk_add_f16"); run<2, 64, 8>("add_u32");
    return 0;
}
