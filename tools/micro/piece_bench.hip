// Cost of the fast kernel's per-piece code in isolation (development aid): fast_count_piece and piece_fail_bits from
// amp_fast.hpp on register data, F_WAVES waves per CU (one block per CU like k_fast), in cycles per piece and wave.
// Build (from the repo root): hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -o /tmp/piece_bench tools/micro/piece_bench.hip
#include "../../amplipy_amd/csrc/amp_fast.hpp"
#include <stdio.h>
using namespace amp;

template <int MODE>
__global__ void __launch_bounds__(F_WAVES * 64, 2) k(uint32_t *out, int tiles, uint32_t seed, unsigned long long *clk) {
    __shared__ uint32_t s_pwin[F_WAVES][F_REP * F_REPW];
    __shared__ uint32_t s_fill[(160256 - F_WAVES * F_REP * F_REPW * 4) / 4 - 64];      // one block per CU, like the kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < F_REP * F_REPW; i += 64) s_pwin[wave][i] = 0;
    if (seed == 12345u) s_fill[threadIdx.x] = 1;                                        // (keeps the array)
    __syncthreads();
    const uint32_t rep = ((uint32_t)lane >> 2) & (uint32_t)(F_REP - 1);
    const uint32_t wrep = (uint32_t)(uintptr_t)((lds_u8 *)s_pwin[wave] + rep * (uint32_t)(F_REPW * 4));
    uint4 q[F_NP]; uint2 s[F_NP];
    uint32_t x = threadIdx.x * 2654435761u + seed;
    for (int k = 0; k < F_NP; ++k) {
        x = x * 1664525u + 1013904223u;
        q[k] = make_uint4(0x25252525u ^ (x & 0x10101010u), 0x25252525u, 0x25250225u, 0x25252525u);
        uint32_t c0 = 0, c1 = 0;
        for (int b = 0; b < 8; ++b) { x = x * 1664525u + 1013904223u; c0 |= (1u << ((x >> 9) & 3u)) << (4 * b); c1 |= (1u << ((x >> 13) & 3u)) << (4 * b); }
        s[k] = make_uint2(c0, c1);
    }
    const uint32_t np = 10, rot = (uint32_t)lane % np;
    const uint32_t mqb = 20u * 0x01010101u;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const int32_t qa = 20 + (lane & 7), qb = 150 - ((lane >> 3) & 7), dbase = 16 + (lane & 3) - qa + (t & 31);
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < F_NP; ++k) {
                uint32_t p = (uint32_t)k + rot;
                p = p >= np ? p - np : p;
                const int32_t j0 = (int32_t)(p * 16u);
                if (fast_count_piece(q[k], s[k], j0, qa, qb, dbase, mqb, (uint32_t)F_PW, wrep)) acc |= 1u << k;
            }
        }
        if (MODE == 1 || MODE == 2) {
            int32_t ffmin = 0x7FFFFFFF, lemax = -1;
#pragma unroll
            for (int k = 0; k < F_NP; ++k) {
                uint32_t p = (uint32_t)k + rot;
                p = p >= np ? p - np : p;
                const int32_t j0 = (int32_t)(p * 16u);
                uint2 nx = make_uint2(q[0].x, q[0].y);
                if (k + 1 < F_NP) nx = make_uint2(q[k + 1].x, q[k + 1].y);
                uint32_t fail = piece_fail_bits<4>(q[k], nx, 80u);
                int32_t blo = qa - j0, bhi = qb - 4 - j0;
                blo = blo < 0 ? 0 : (blo > 16 ? 16 : blo); bhi = bhi > 15 ? 15 : (bhi < -1 ? -1 : bhi);
                fail &= (0xFFFFu >> (15 - bhi)) & (0xFFFFu << blo);
                const int32_t f1 = j0 + (__builtin_ffs((int)fail) - 1), e1 = j0 + (31 - __builtin_clz(fail)) + 4;
                ffmin = fail && f1 < ffmin ? f1 : ffmin;
                lemax = fail && e1 > lemax ? e1 : lemax;
            }
            acc += (uint32_t)(ffmin + lemax);
            q[t % F_NP].y ^= acc & 0x01000000u;           // (the scan stays in the loop)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int i = lane; i < F_REP * F_REPW; i += 64) acc += s_pwin[wave][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char *name) {
    uint32_t *d; unsigned long long *clk, h;
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tiles = 2000;
    k<MODE><<<256, F_WAVES * 64>>>(d, 10, 1, clk);
    (void)hipEventRecord(e0);
    k<MODE><<<256, F_WAVES * 64>>>(d, tiles, 2, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-14s %.3f ms for %d tiles of %d pieces per wave: %.2f us per tile, %.0f shader cycles per piece and wave\n", name, ms, tiles, F_NP,
           ms * 1e3 / tiles, (double)h / tiles / F_NP);
    (void)hipFree(d); (void)hipFree(clk);
}
int main() {
    run<0>("count"); run<1>("scan"); run<2>("count + scan");
    return 0;
}
