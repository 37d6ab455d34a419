// Issue cost of the instructions the scan kernels are made of, gfx950 (development aid): every kernel is a long unrolled
// run of ONE instruction on independent registers; printed: cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/vr3 tools/micro/valu_rate3.hip && /tmp/vr3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define BODY(ASM) \
    for (int it = 0; it < iters; ++it) { \
        asm volatile(REP8(REP8(ASM)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(q0), "+v"(q1) : "v"(k0), "v"(k1), "s"(sk) : "vcc", "memory"); \
    }

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b * 3u;
    uint32_t e = a ^ 0x1111u, f = b ^ 0x2222u, g = c ^ 0x3333u, h = d ^ 0x4444u;
    uint64_t q0 = ((uint64_t)a << 32) | b, q1 = ((uint64_t)c << 32) | d;
    uint32_t k0 = seed | 0x01010101u, k1 = (threadIdx.x * 16u) & 0x3FF0u, sk = seed;
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = 0;
    __syncthreads();
    if (KIND == 0) BODY("v_add_u32 %0, %0, %10\n")
    if (KIND == 1) BODY("v_and_or_b32 %0, %0, %10, %11\n")
    if (KIND == 2) BODY("v_bitop3_b32 %0, %0, %10, %11 bitop3:0x6c\n")
    if (KIND == 3) BODY("v_perm_b32 %0, %0, %10, %11\n")
    if (KIND == 4) BODY("v_dot4_u32_u8 %0, %1, %10, %0\n")
    if (KIND == 5) BODY("v_qsad_pk_u16_u8 %8, %9, %10, %8\n")
    if (KIND == 6) BODY("v_lshlrev_b32_sdwa %0, %10, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n")
    if (KIND == 7) BODY("v_bfm_b32 %0, %0, %10\n")
    if (KIND == 8) BODY("v_med3_i32 %0, %0, %10, %11\n")
    if (KIND == 9) BODY("v_bcnt_u32_b32 %0, %1, %0\n")
    if (KIND == 10) BODY("v_ffbl_b32 %0, %0\n")
    if (KIND == 11) BODY("v_pk_lshrrev_b16 %0, 15, %0\n")
    if (KIND == 12) BODY("v_mul_u32_u24 %0, %0, %10\n")
    if (KIND == 13) BODY("v_cmp_lt_u32 vcc, %0, %10\nv_cndmask_b32 %0, %0, %10, vcc\n")
    if (KIND == 14) BODY("v_alignbit_b32 %0, %0, %10, 8\n")
    if (KIND == 15) BODY("v_lshlrev_b64 %8, 3, %8\n")
    if (KIND == 16) BODY("v_mul_lo_u32 %0, %0, %10\n")
    if (KIND == 17) BODY("v_add3_u32 %0, %0, %10, %11\n")
    if (KIND == 18) BODY("v_lshl_or_b32 %0, %0, 3, %10\n")
    if (KIND == 19) BODY("ds_add_u32 %11, %10\n")
    if (KIND == 20) BODY("ds_read_b64 %8, %11\ns_waitcnt lgkmcnt(0)\n")     // not meaningful as issue cost (waits) -- latency per dependent read
    if (KIND == 21) BODY("v_sad_u8 %0, %0, %10, %11\n")
    if (KIND == 22) BODY("v_min_u32 %0, %0, %10\n")
    if (KIND == 23) BODY("v_mad_u32_u24 %0, %0, %10, %11\n")
    if (KIND == 24) BODY("s_and_b32 %12, %12, 7\n")                         // scalar: issue cost beside nothing
    if (KIND == 25) BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n")
    if (KIND == 26) BODY("v_lshrrev_b32 %0, 3, %0\n")
    if (KIND == 27) BODY("v_sub_u32 %0, %0, %10 clamp\n")
    if (KIND == 28) BODY("v_readlane_b32 s20, %0, 3\n")
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)q0 ^ (uint32_t)q1 ^ lds[threadIdx.x];
}

template <int KIND>
static void run(const char *name, int per_asm) {
    uint32_t *d; (void)hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 256;
    printf("%-26s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;          // 4 waves per block -> one wave per SIMD per block
        k<KIND><<<blocks, 256>>>(d, 4, 1);
        (void)hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(d, iters, 2);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 64 * per_asm * wps;       // per SIMD
        printf("  %dw/SIMD %.2f ns = %.2f cyc@2.2GHz", wps, ms * 1e6 / insts, ms * 1e6 / insts * 2.2);
    }
    printf("\n");
    (void)hipFree(d);
}

int main() {
    run<0>("v_add_u32", 1); run<1>("v_and_or_b32", 1); run<2>("v_bitop3_b32", 1); run<3>("v_perm_b32", 1); run<4>("v_dot4_u32_u8", 1);
    run<5>("v_qsad_pk_u16_u8", 1); run<6>("v_lshlrev_b32_sdwa", 1); run<7>("v_bfm_b32", 1); run<8>("v_med3_i32", 1); run<9>("v_bcnt_u32_b32", 1);
    run<10>("v_ffbl_b32", 1); run<11>("v_pk_lshrrev_b16", 1); run<12>("v_mul_u32_u24", 1); run<13>("v_cmp+v_cndmask (2)", 2); run<14>("v_alignbit_b32", 1);
    run<15>("v_lshlrev_b64", 1); run<16>("v_mul_lo_u32", 1); run<17>("v_add3_u32", 1); run<18>("v_lshl_or_b32", 1); run<19>("ds_add_u32 (no conflict)", 1);
    run<20>("ds_read_b64+wait", 1); run<21>("v_sad_u8", 1); run<22>("v_min_u32", 1); run<23>("v_mad_u32_u24", 1); run<24>("s_and_b32", 1);
    run<25>("v_mov_b32_dpp", 1); run<26>("v_lshrrev_b32", 1); run<27>("v_sub_u32 clamp", 1); run<28>("v_readlane_b32", 1);
    return 0;
}
