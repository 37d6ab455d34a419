// LDS atomic cost of the counting pass on a PILE of reads (development aid): 8 waves per CU, every wave adds 16 bases of
// piece (k + lane) mod 10 of its lane's read at window offset start(lane) + 16 piece, for replica / stride layouts:
//   0: 4 replicas of 225 words, replica (lane >> 2) & 3           (k_fast6 as first written)
//   1: 4 replicas of 232 words (stride = 8 banks), replica (lane >> 1) & 3
//   2: 8 replicas of 257 words, replica (lane >> 2) & 7           (k_fast)
//   3: conflict-free reference: lane l adds to word l of a 64-word row
//   4: 4 replicas of 228 words (stride = 4 banks), replica (lane >> 1) & 3
//   5: 4 replicas of 240 words (stride = 16 banks), replica lane & 3
// hipcc -O3 --offload-arch=gfx950 -o tools/micro/bin/lds_pile tools/micro/lds_pile.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int MODE>
__global__ void __launch_bounds__(512) k(uint32_t *out, int iters, unsigned long long *cyc, int spread) {
    __shared__ uint32_t win[8][2100];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 2100; i += 64) win[wave][i] = 0;
    __syncthreads();
    const uint32_t wbase = (uint32_t)(uintptr_t)(lds_u32 *)win[wave];
    // a sorted tile of a pile: `spread` distinct start offsets, equal starts side by side
    const uint32_t start = 16u + ((uint32_t)lane * (uint32_t)spread) / 64u;
    uint32_t rep, rs;
    if (MODE == 0) { rep = (lane >> 2) & 3; rs = 225; }
    else if (MODE == 1) { rep = (lane >> 1) & 3; rs = 232; }
    else if (MODE == 2) { rep = (lane >> 2) & 7; rs = 257; }
    else if (MODE == 4) { rep = (lane >> 1) & 3; rs = 228; }
    else if (MODE == 5) { rep = lane & 3; rs = 240; }
    else { rep = 0; rs = 0; }
    const uint32_t rot = (uint32_t)lane % 10u;
    uint32_t val = 1u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 10; ++kk) {
            uint32_t p = (uint32_t)kk + rot; p = p >= 10u ? p - 10u : p;
            uint32_t wb = MODE == 3 ? wbase + 4u * (uint32_t)lane : wbase + 4u * (rep * rs + start + 16u * p);
#define A(B) asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"(wb), "v"(val), "n"(MODE == 3 ? 0 : 4 * B) : "memory");
            A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(12) A(13) A(14) A(15)
#undef A
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    uint32_t s = 0;
    for (int i = lane; i < 2100; i += 64) s += win[wave][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int spread) {
    uint32_t *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 200;
    k<MODE><<<256, 512>>>(out, 5, cyc, spread);
    k<MODE><<<256, 512>>>(out, iters, cyc, spread);
    (void)hipDeviceSynchronize();
    unsigned long long c[256]; (void)hipMemcpy(c, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += (double)c[i]; m /= 256;
    printf("%-52s spread %2d: %.2f cycles per ds_add wave-instruction per CU\n", name, spread, m / ((double)iters * 160 * 8));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int spread : {1, 7, 13, 40}) {
        run<0>("0: 4 x 225, replica (lane >> 2) & 3", spread);
        run<1>("1: 4 x 232, replica (lane >> 1) & 3", spread);
        run<4>("4: 4 x 228, replica (lane >> 1) & 3", spread);
        run<5>("5: 4 x 240, replica lane & 3", spread);
        run<2>("2: 8 x 257, replica (lane >> 2) & 7", spread);
        run<3>("3: conflict-free", spread);
    }
    return 0;
}
