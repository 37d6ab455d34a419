// Issue rate of individual VALU instructions on gfx950, 4 waves per SIMD (development aid): inline asm, 8 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHAIN8(INS) \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c), "s"(sc) : "vcc", "s20", "s21");
#define I_ADD(n)   "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_AND(n)   "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_XOR(n)   "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_LSHL(n)  "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_BFE(n)   "v_bfe_u32 %" #n ", %" #n ", 3, 29\n"
#define I_FMA(n)   "v_fma_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_ADDF(n)  "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %" #n "\n"
#define I_ADD3(n)  "v_add3_u32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_PERM(n)  "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 8\n"
#define I_SDWA(n)  "v_lshlrev_b32_sdwa %" #n ", %8, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
#define I_DOT4(n)  "v_dot4_u32_u8 %" #n ", %" #n ", %8, %" #n "\n"
#define I_CNDM(n)  "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_MOV(n)   "v_mov_b32 %" #n ", %8\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define I_OR(n)    "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_SUB(n)   "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_LSHR(n)  "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define I_BITOP(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %" #n " bitop3:0x80\n"
#define I_CMP(n)   "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_OR3(n)   "v_or3_b32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_XAD(n)   "v_xad_u32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MAX(n)   "v_max_u32 %" #n ", %" #n ", %8\n"
#define I_ADDCO(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_BCNT(n)  "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %" #n ", %8\n"
#define I_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define I_MIN(n)   "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_MED3(n)  "v_med3_i32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_BFI(n)   "v_bfi_b32 %" #n ", %8, %" #n ", %" #n "\n"

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t r[8];
    for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * 2654435761u + seed * (j + 1);
    uint32_t c = seed | 1u, sc = 0x07030602u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) CHAIN8(I_ADD) if (KIND == 1) CHAIN8(I_AND) if (KIND == 2) CHAIN8(I_XOR) if (KIND == 3) CHAIN8(I_LSHL)
            if (KIND == 4) CHAIN8(I_BFE) if (KIND == 5) CHAIN8(I_FMA) if (KIND == 6) CHAIN8(I_ADDF) if (KIND == 7) CHAIN8(I_PKADD)
            if (KIND == 8) CHAIN8(I_MAD24) if (KIND == 9) CHAIN8(I_ADD3) if (KIND == 10) CHAIN8(I_PERM) if (KIND == 11) CHAIN8(I_ALIGN)
            if (KIND == 12) CHAIN8(I_SDWA) if (KIND == 13) CHAIN8(I_DOT4) if (KIND == 14) CHAIN8(I_CNDM) if (KIND == 15) CHAIN8(I_MOV)
            if (KIND == 16) CHAIN8(I_ANDOR) if (KIND == 17) CHAIN8(I_LSHLADD) if (KIND == 18) CHAIN8(I_BCNT) if (KIND == 19) CHAIN8(I_PKSUB)
            if (KIND == 20) CHAIN8(I_MIN) if (KIND == 21) CHAIN8(I_MED3) if (KIND == 22) CHAIN8(I_MULLO) if (KIND == 23) CHAIN8(I_BFI)
            if (KIND == 24) CHAIN8(I_OR) if (KIND == 25) CHAIN8(I_SUB) if (KIND == 26) CHAIN8(I_LSHR) if (KIND == 27) CHAIN8(I_BITOP)
            if (KIND == 28) CHAIN8(I_CMP) if (KIND == 29) CHAIN8(I_OR3) if (KIND == 30) CHAIN8(I_XAD) if (KIND == 31) CHAIN8(I_MAX) if (KIND == 32) CHAIN8(I_ADDCO)
        }
    }
    uint32_t x = 0;
    for (int j = 0; j < 8; ++j) x ^= r[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int KIND> static void run(const char *name) {
    uint32_t *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2048, wps = 4, blocks = 256 * wps;
    k<KIND><<<blocks, 256>>>(d, 16, 1);
    (void)hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters, 2);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 64 * wps;
    printf("%-16s %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e6 / insts * 2.4);
    (void)hipFree(d);
}
int main() {
    run<0>("v_add_u32"); run<1>("v_and_b32"); run<2>("v_xor_b32"); run<3>("v_lshlrev_b32"); run<4>("v_bfe_u32"); run<5>("v_fma_f32");
    run<6>("v_add_f32"); run<7>("v_pk_add_u16"); run<8>("v_mad_u32_u24"); run<9>("v_add3_u32"); run<10>("v_perm_b32"); run<11>("v_alignbit_b32");
    run<12>("v_lshlrev_sdwa"); run<13>("v_dot4_u32_u8"); run<14>("v_cndmask_b32"); run<15>("v_mov_b32"); run<16>("v_and_or_b32");
    run<17>("v_lshl_add_u32"); run<18>("v_bcnt_u32_b32"); run<19>("v_pk_sub_i16"); run<20>("v_min_u32"); run<21>("v_med3_i32");
    run<22>("v_mul_lo_u32"); run<23>("v_bfi_b32"); run<24>("v_or_b32"); run<25>("v_sub_u32"); run<26>("v_lshrrev_b32"); run<27>("v_bitop3_b32");
    run<28>("v_cmp_lt_u32"); run<29>("v_or3_b32"); run<30>("v_xad_u32"); run<31>("v_max_u32"); run<32>("v_mul_u32_u24");
    return 0;
}
