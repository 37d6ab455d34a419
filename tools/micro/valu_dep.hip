// VALU issue rate on gfx950 against occupancy and dependency (development aid): W waves per SIMD, every wave runs a
// chain of v_add_u32 in which each instruction depends on the one D instructions earlier (D = 1: fully dependent,
// D = 8: eight independent chains).  Also prints the shader clock seen by s_memtime against the wall clock.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_dep valu_dep.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int D>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed, unsigned long long *clk) {
    uint32_t r[8];
    for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * 2654435761u + seed * (j + 1);
    const uint32_t c = seed | 1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (D == 8)
                asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c));
            else if (D == 2)
                asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\n"
                             "v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c));
            else
                asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %0, %0, %8\n"
                             "v_add_u32 %0, %0, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %0, %0, %8\nv_add_u32 %0, %0, %8\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    uint32_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= r[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int D> static void run(int wps) {
    uint32_t *d; unsigned long long *clk, h[2];
    (void)hipMalloc(&d, 256 * 4096 * 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * wps;            // blocks of 4 waves: one wave per SIMD and block
    k<D><<<blocks, 256>>>(d, 16, 1, clk);
    (void)hipEventRecord(e0);
    k<D><<<blocks, 256>>>(d, iters, 2, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double insts = (double)iters * 64 * wps;           // per SIMD
    printf("waves/SIMD %d, dependency distance %d: %.3f ms -> %.2f ns per wave-instruction per SIMD; s_memtime %.0f ticks per 100 MHz wall tick x 100 = %.0f MHz\n",
           wps, D, ms, ms * 1e6 / insts, (double)h[0] / (double)h[1], 100.0 * (double)h[0] / (double)h[1]);
    (void)hipFree(d); (void)hipFree(clk);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<8>(w); }
    return 0;
}
