#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int J> __device__ __forceinline__ uint32_t shl_byte(uint32_t sh, uint32_t val) {
    uint32_t r;
    if (J == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(val));
    if (J == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(val));
    if (J == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(val));
    if (J == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(val));
    return r;
}
__global__ void k(uint32_t *out) {
    const uint32_t sh = 0x18100800u + threadIdx.x * 0u, val = 0x01000101u;   // shifts 0,8,16,24 ; values 1,1,0,1
    out[threadIdx.x * 4 + 0] = shl_byte<0>(sh, val);
    out[threadIdx.x * 4 + 1] = shl_byte<1>(sh, val);
    out[threadIdx.x * 4 + 2] = shl_byte<2>(sh, val);
    out[threadIdx.x * 4 + 3] = shl_byte<3>(sh | 0xE0000000u, val);   // junk above bit 4 of the shift byte must be ignored
    uint32_t d = __builtin_amdgcn_udot4(0x80008080u, 0x08040201u, 0u, false);
    if (threadIdx.x == 0) out[256] = d;
}
int main() {
    uint32_t *d, h[260]; hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%08x %08x %08x %08x dot4 %u (expect 00000001 00000100 00000000 01000000 dot4 %u)\n", h[0], h[1], h[2], h[3], h[256], 128u * (1 + 2 + 8));
    return 0;
}
