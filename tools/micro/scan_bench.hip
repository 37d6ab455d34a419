// What does the sliding-window scan (k_tile's P2) cost as a kernel of its own, at higher occupancy?
// (development aid for DESIGN.md section 4, "plan for the next round").  Same device code (amp_tile.hpp
// p2_round), same synthetic quality distribution, one tile of 64 reads per wave, 4 waves per block,
// per-wave LDS = chunk map 2 KB + state words 1.5 KB.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o /tmp/scan_bench tools/micro/scan_bench.hip && /tmp/scan_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../amplipy_amd/csrc/amp_read.hpp"
#include "../../amplipy_amd/csrc/amp_tile.hpp"

using namespace amp;

template <int WAVES, int MINB>
__global__ void __launch_bounds__(WAVES * 64, MINB)
k_scan_only(const uint8_t *qual, const uint32_t *off8, const uint32_t *lseq, const uint16_t *flag, int64_t n, int32_t mq, uint32_t *iq_out) {
    __shared__ uint32_t s_st[WAVES][S_WORDS * TILE];
    __shared__ uint32_t s_map[WAVES][T_MAPCAP / 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_u32 *const st = (lds_u32 *)s_st[wave];
    lds_u8 *const cmap = (lds_u8 *)s_map[wave];
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t i = tile * TILE + lane;
        const bool valid = i < n;
        const int32_t L = valid ? (int32_t)lseq[i] : 0;
        const uint32_t o8 = valid ? off8[i] : 0u;
        const bool rev = valid && (flag[i] & 0x10u);
        const int32_t lo = 0, hi = L, Wd = 4;
        uint32_t nch2 = (valid && L >= Wd) ? (uint32_t)(((hi - Wd) >> 3) - (lo >> 3) + 1) : 0u;
        uint32_t total2;
        const uint32_t cb2 = wave_excl_scan(nch2, lane, total2);
        st[S_OFF8 * TILE + lane] = o8;
        st[S_LOHI * TILE + lane] = (uint32_t)lo | ((uint32_t)hi << 16);
        st[S_FF * TILE + lane] = 0xFFFFu;
        st[S_REV * TILE + lane] = rev ? 1u : 0u;
        st[S_CB2 * TILE + lane] = cb2;
        const ChunkEnv env{cmap, st, nullptr, nullptr, nullptr, qual, nullptr, nullptr, 0, 0u, 0u, mq};
        for (uint32_t base = 0; base < total2; base += T_MAPCAP) {
            wave_sync();
            {
                uint32_t a = cb2 > base ? cb2 : base, b = cb2 + nch2 < base + T_MAPCAP ? cb2 + nch2 : base + T_MAPCAP;
                for (uint32_t c = a; c < b; ++c) cmap[c - base] = (uint8_t)lane;
            }
            wave_sync();
            const uint32_t lim = total2 - base < (uint32_t)T_MAPCAP ? total2 - base : (uint32_t)T_MAPCAP;
            p2_round<4>(env, lane, lim, base, (uint32_t)mq * 4u);
        }
        wave_sync();
        if (valid) iq_out[i] = st[S_FF * TILE + lane];
    }
}

int main() {
    const int64_t n = 1993533; const int L = 150, stride = 152;
    std::vector<uint8_t> q((size_t)n * stride + 64, 0);
    std::vector<uint32_t> off8(n), lseq(n, L); std::vector<uint16_t> flag(n);
    srand(1);
    for (int64_t i = 0; i < n; ++i) {
        off8[i] = (uint32_t)(i * stride / 8); flag[i] = (rand() & 1) ? 16 : 0;
        for (int k = 0; k < L; ++k) { int r = rand() % 100; q[(size_t)i * stride + k] = r < 80 ? 37 : r < 92 ? 25 : r < 98 ? 11 : 2; }
    }
    uint8_t *dq; uint32_t *doff, *dl, *dout; uint16_t *df;
    hipMalloc(&dq, q.size()); hipMalloc(&doff, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&df, n * 2); hipMalloc(&dout, n * 4);
    hipMemcpy(dq, q.data(), q.size(), hipMemcpyHostToDevice); hipMemcpy(doff, off8.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dl, lseq.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(df, flag.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int threads, int grid) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0);
            kern<<<grid, threads>>>(dq, doff, dl, df, n, 20, dout);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 1 && ms < best) best = ms;
        }
        printf("%-40s grid %6d x %4d threads: %.1f us  (%.2f TB/s of qualities)\n", name, grid, threads, best * 1e3, (double)n * stride / (best * 1e-3) / 1e12);
    };
    const int tiles = (int)((n + 63) / 64);
    run("4 waves/block, min 2 blocks/CU", k_scan_only<4, 2>, 256, tiles / 4 + 1);
    run("4 waves/block, min 4 blocks/CU", k_scan_only<4, 4>, 256, tiles / 4 + 1);
    run("4 waves/block, min 6 blocks/CU", k_scan_only<4, 6>, 256, tiles / 4 + 1);
    run("4 waves/block, min 8 blocks/CU", k_scan_only<4, 8>, 256, tiles / 4 + 1);
    run("4 waves/block, min 8, persistent x2048", k_scan_only<4, 8>, 256, 2048);
    std::vector<uint32_t> out(n);
    hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    unsigned long long acc = 0; for (auto v : out) acc += v;
    printf("checksum %llu\n", acc);
    return 0;
}
