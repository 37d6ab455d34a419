// LDS-DMA fill patterns for a tile of 64 reads (152 quality bytes + 76 packed-base bytes per read), gfx950 (development aid).
//   A: the tile's bytes as ONE contiguous run: instruction s, lane l moves run[1024 s + 16 l, +16)
//   B: per-lane rows: instruction s, lane l moves bytes [16 s, +16) of ITS OWN row; LDS image piece-major (piece s of lane l
//      at 1024 s + 16 l): no address arithmetic per instruction (immediate offsets), conflict-free 16-byte reads
// Prints GB/s and a checksum of everything read back from LDS (equal for A and B = both images are right).
// hipcc -O3 --offload-arch=gfx950 -o tools/micro/bin/dma tools/micro/dma_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ void dma16(const void *g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory", "m0");
}
template <int OFF>
__device__ __forceinline__ void dma16_off(const void *g, uint32_t lds_addr) {      // global address + OFF, LDS address m0 + OFF + 16 lane
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(g), "s"(lds_addr), "n"(OFF) : "memory", "m0");
}

constexpr int WAVES = 8, QROW = 152, SROW = 76, QT = 10240, ST = 5120;

template <int PATTERN>
__global__ void __launch_bounds__(WAVES * 64) k(const uint8_t *qual, const uint8_t *seq, int64_t n_tiles, uint32_t *out) {
    __shared__ uint4 s_q[WAVES][QT / 16];
    __shared__ uint4 s_s[WAVES][ST / 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u8 *)s_q[wave]), sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u8 *)s_s[wave]);
    uint32_t acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * WAVES + wave; t < n_tiles; t += (int64_t)gridDim.x * WAVES) {
        const uint8_t *qrun = qual + t * 64 * QROW, *srun = seq + t * 64 * SROW;
        if (PATTERN == 0) {
#pragma unroll
            for (int s = 0; s < 10; ++s) { uint32_t off = s * 1024 + lane * 16; off = off < 64 * QROW - 16 ? off : 64 * QROW - 16; dma16(qrun + off, qb + s * 1024); }
#pragma unroll
            for (int s = 0; s < 5; ++s) { uint32_t off = s * 1024 + lane * 16; off = off < 64 * SROW - 16 ? off : 64 * SROW - 16; dma16(srun + off, sb + s * 1024); }
        } else if (PATTERN == 2) {
            // B': instruction s, lane l moves chunk 2 (s >> 1) + (l & 1) of row 32 (s & 1) + (l >> 1): 32 contiguous bytes per row
            const uint8_t *qe = qrun + (lane >> 1) * QROW + (lane & 1) * 16, *qo = qe + 32 * QROW, *srow = srun + lane * SROW;
            dma16_off<0>(qe, qb); dma16_off<0>(qo, qb + 1024); dma16_off<32>(qe, qb + 2048 - 32); dma16_off<32>(qo, qb + 3072 - 32); dma16_off<64>(qe, qb + 4096 - 64);
            dma16_off<64>(qo, qb + 5120 - 64); dma16_off<96>(qe, qb + 6144 - 96); dma16_off<96>(qo, qb + 7168 - 96); dma16_off<128>(qe, qb + 8192 - 128); dma16_off<128>(qo, qb + 9216 - 128);
            dma16_off<0>(srow, sb); dma16_off<16>(srow, sb + 1008 * 1); dma16_off<32>(srow, sb + 1008 * 2); dma16_off<48>(srow, sb + 1008 * 3); dma16_off<64>(srow, sb + 1008 * 4);
        } else {
            const uint8_t *qrow = qrun + lane * QROW, *srow = srun + lane * SROW;
            dma16_off<0>(qrow, qb); dma16_off<16>(qrow, qb + 1008 * 1); dma16_off<32>(qrow, qb + 1008 * 2); dma16_off<48>(qrow, qb + 1008 * 3); dma16_off<64>(qrow, qb + 1008 * 4);
            dma16_off<80>(qrow, qb + 1008 * 5); dma16_off<96>(qrow, qb + 1008 * 6); dma16_off<112>(qrow, qb + 1008 * 7); dma16_off<128>(qrow, qb + 1008 * 8); dma16_off<144>(qrow, qb + 1008 * 9);
            dma16_off<0>(srow, sb); dma16_off<16>(srow, sb + 1008 * 1); dma16_off<32>(srow, sb + 1008 * 2); dma16_off<48>(srow, sb + 1008 * 3); dma16_off<64>(srow, sb + 1008 * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // read the lane's own row back (38 + 19 dwords) and fold it into a checksum
        if (PATTERN == 0) {
            const lds_u32 *q = (const lds_u32 *)((lds_u8 *)s_q[wave] + lane * QROW), *s = (const lds_u32 *)((lds_u8 *)s_s[wave] + lane * SROW);
#pragma unroll
            for (int j = 0; j < 38; ++j) acc += q[j] * (uint32_t)(j + 1);
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += s[j] * (uint32_t)(j + 101);
        } else if (PATTERN == 2) {
            const lds_u32 *q = (const lds_u32 *)((lds_u8 *)s_q[wave] + 1024 * (lane >> 5) + 32 * (lane & 31)), *s = (const lds_u32 *)((lds_u8 *)s_s[wave] + lane * 16);
#pragma unroll
            for (int j = 0; j < 38; ++j) { const int pc = j >> 2; acc += q[(2048 * (pc >> 1) + 16 * (pc & 1)) / 4 + (j & 3)] * (uint32_t)(j + 1); }
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += s[(j >> 2) * 256 + (j & 3)] * (uint32_t)(j + 101);
        } else {
            const lds_u32 *q = (const lds_u32 *)((lds_u8 *)s_q[wave] + lane * 16), *s = (const lds_u32 *)((lds_u8 *)s_s[wave] + lane * 16);
#pragma unroll
            for (int j = 0; j < 38; ++j) acc += q[(j >> 2) * 256 + (j & 3)] * (uint32_t)(j + 1);
#pragma unroll
            for (int j = 0; j < 19; ++j) acc += s[(j >> 2) * 256 + (j & 3)] * (uint32_t)(j + 101);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    atomicAdd(out, acc);
}

int main() {
    const int64_t n_tiles = 31149, n = n_tiles * 64;
    uint8_t *q, *s; uint32_t *out;
    (void)hipMalloc(&q, n * QROW + 4096); (void)hipMalloc(&s, n * SROW + 4096); (void)hipMalloc(&out, 4);
    std::vector<uint8_t> h(n * QROW + 4096);
    uint32_t x = 12345u;
    for (auto &b : h) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); }
    (void)hipMemcpy(q, h.data(), n * QROW + 4096, hipMemcpyHostToDevice);
    (void)hipMemcpy(s, h.data() + 777, n * SROW + 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)n * (QROW + SROW);
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 3; ++pat) {
            (void)hipMemset(out, 0, 4);
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                (void)hipMemset(out, 0, 4);
                (void)hipEventRecord(e0);
                if (pat == 0) k<0><<<256, WAVES * 64>>>(q, s, n_tiles, out); else if (pat == 1) k<1><<<256, WAVES * 64>>>(q, s, n_tiles, out); else k<2><<<256, WAVES * 64>>>(q, s, n_tiles, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            uint32_t hs; (void)hipMemcpy(&hs, out, 4, hipMemcpyDeviceToHost);
            printf("pattern %c: %.3f ms for %.1f MB -> %.2f TB/s; checksum %08x (%s)\n", "ABC"[pat], best, bytes / 1e6, bytes / best / 1e9, hs, hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
