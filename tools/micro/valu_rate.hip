// Integer VALU issue rate on gfx950 (development aid): cycles per wave64 instruction per SIMD for a few op kinds
// at 1, 2, 4 and 8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/vr tools/micro/valu_rate.hip && /tmp/vr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b * 3u;
    uint32_t e = a ^ 0x1111u, f = b ^ 0x2222u, g = c ^ 0x3333u, h = d ^ 0x4444u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { a = (a + b) ^ c; e = (e + f) ^ g; b = (b + c) ^ d; f = (f + g) ^ h; }          // add, xor
            if (KIND == 1) { a = (a >> 3) & 0x0F0F0F0Fu; e = (e >> 5) & 0x33333333u; b = b << 1 | (a & 1); f = f << 1 | (e & 1); }
            if (KIND == 2) { a = __builtin_amdgcn_perm(a, b, 0x05010400u); e = __builtin_amdgcn_perm(e, f, 0x07030602u); b = __builtin_amdgcn_alignbit(b, a, 8); f = __builtin_amdgcn_alignbit(f, e, 16); }
            if (KIND == 3) { a = __builtin_amdgcn_ubfe(a + b, 3, 9); e = __builtin_amdgcn_ubfe(e + f, 5, 11); b += a; f += e; }
            if (KIND == 4) { a = (a & 0xFFFFu) * (b & 0xFFu) + c; e = (e & 0xFFFFu) * (f & 0xFFu) + g; }                         // mad_u32_u24
            if (KIND == 5) { uint64_t x = ((uint64_t)a << 32 | b) >> (c & 31); a = (uint32_t)x; uint64_t y = ((uint64_t)e << 32 | f) >> (g & 31); e = (uint32_t)y; }
            if (KIND == 6) { uint64_t x = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)a | ((uint64_t)b << 32), c, 0ull); a ^= (uint32_t)x; b ^= (uint32_t)(x >> 32); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

template <int KIND>
static void run(const char *name, int insts_per_unroll) {
    uint32_t *d; hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;          // 4 waves per block -> one wave per SIMD per block
        k<KIND><<<blocks, 256>>>(d, 16, 1);
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(d, iters, 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 16 * insts_per_unroll * wps;       // per SIMD
        printf("%-18s %d waves/SIMD: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms,
               ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    }
    hipFree(d);
}

int main() {
    run<0>("add/xor", 8); run<1>("shift/and/or", 8); run<2>("perm/alignbit", 4); run<3>("add+bfe", 6);
    run<4>("and,and,mad_u24", 6); run<5>("lshr_b64", 2); run<6>("qsad+xor+xor", 3);
    return 0;
}
