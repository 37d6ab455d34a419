// Microbenchmark (development aid): LDS atomic-add throughput on gfx950 for the access
// patterns of the pileup kernel.  Build: hipcc -O3 --offload-arch=gfx950 -o lds_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t win[6 * 1024];
    lds_u32 *w = (lds_u32 *)win;
    for (int i = threadIdx.x; i < 6 * 1024; i += blockDim.x) w[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = threadIdx.x * 2654435761u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint32_t idx;
            if (MODE == 0) idx = (uint32_t)lane + 64u * (uint32_t)((it + b) & 7);                  // distinct consecutive addresses
            else if (MODE == 1) idx = (uint32_t)(it & 1023);                                     // all lanes same address
            else if (MODE == 2) { x = x * 1664525u + 1013904223u; idx = (x >> 10) & 1023u; idx += ((x >> 3) & 3u) * 1024u; }  // random
            else if (MODE == 3) {   // kernel-like: lane k -> chunk k of reads of 19 chunks, rotated base, random plane
                x = x * 1664525u + 1013904223u;
                uint32_t rot = ((uint32_t)lane >> 2) & 7u;
                uint32_t chunk = (uint32_t)lane % 19u, rdi = (uint32_t)lane / 19u;
                idx = ((x >> 5) & 3u) * 1024u + ((wave * 7 + rdi * 2 + chunk * 8 + ((rot + b) & 7u) + (it & 63)) & 1023u);
            } else {               // kernel-like without rotation
                x = x * 1664525u + 1013904223u;
                uint32_t chunk = (uint32_t)lane % 19u, rdi = (uint32_t)lane / 19u;
                idx = ((x >> 5) & 3u) * 1024u + ((wave * 7 + rdi * 2 + chunk * 8 + b + (it & 63)) & 1023u);
            }
            __hip_atomic_fetch_add(w + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < 6 * 1024; i += blockDim.x) s += w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int blocks, int threads) {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    double wave_instrs_per_cu = (double)iters * 8 * (threads / 64);
    printf("%-28s threads=%4d: %.3f ms, block0 %.0f ticks -> %.1f ticks per wave-instr per CU (%.2f G lane-atomics/s chip)\n", name, threads, ms,
           (double)c0, (double)c0 / wave_instrs_per_cu, blocks * wave_instrs_per_cu * 64 / ms / 1e6);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 1024}) {
        run<0>("distinct consecutive", 256, th);
        run<1>("same address", 256, th);
        run<2>("random", 256, th);
        run<3>("kernel-like rotated", 256, th);
        run<4>("kernel-like unrotated", 256, th);
    }
    return 0;
}
