// Checks the byte-parallel helpers of amp_fast.hpp against per-base loops on the GPU (development aid).
//   hipcc --offload-arch=gfx950 -O2 -I. -o /tmp/fhc tools/micro/fast_helpers_check.hip && /tmp/fhc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../amplipy_amd/csrc/amp_fast.hpp"
using namespace amp;

__global__ void k_check(uint32_t seed, unsigned long long *bad) {
    uint32_t x = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (int it = 0; it < 256; ++it) {
        // spread_codes / not_acgt / col_bytes
        uint2 s = make_uint2(rnd(), rnd());
        if (it & 1) { s.x &= 0x84218421u | rnd(); }
        uint32_t cb[4];
        spread_codes(s, cb);
        for (int b = 0; b < 16; ++b) {
            const uint32_t byte = ((b < 8 ? s.x : s.y) >> (8 * ((b & 7) >> 1))) & 0xFFu;
            const uint32_t code = (b & 1) ? (byte & 15u) : (byte >> 4);
            const uint32_t got = (cb[b >> 2] >> (8 * (b & 3))) & 0xFFu;
            if (got != code) atomicAdd(&bad[0], 1ull);
            const bool acgt = code == 1 || code == 2 || code == 4 || code == 8;
            const uint32_t na = (not_acgt(cb[b >> 2]) >> (8 * (b & 3) + 7)) & 1u;
            if (na != (acgt ? 0u : 1u)) atomicAdd(&bad[1], 1ull);
            const uint32_t col = (col_bytes(cb[b >> 2]) >> (8 * (b & 3))) & 0xFFu;
            if (acgt && col != (code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u)) atomicAdd(&bad[2], 1ull);
            if (col > 3u) atomicAdd(&bad[2], 1ull);
        }
        // nibble_to_bytes
        const uint32_t m = rnd() & 0xFFFFu;
        for (int d = 0; d < 4; ++d) {
            const uint32_t nb = nibble_to_bytes(m, d);
            for (int i = 0; i < 4; ++i)
                if (((nb >> (8 * i)) & 0xFFu) != ((m >> (4 * d + i)) & 1u)) atomicAdd(&bad[3], 1ull);
        }
        // ok_bits4
        const uint32_t q = rnd(), mq = rnd() % 129u;
        const uint32_t ok = ok_bits4(q, mq * 0x01010101u);
        for (int i = 0; i < 4; ++i)
            if (((ok >> (8 * i + 7)) & 1u) != ((((q >> (8 * i)) & 0xFFu) >= mq) ? 1u : 0u) || (ok & 0x7F7F7F7Fu)) atomicAdd(&bad[4], 1ull);
    }
}

int main() {
    unsigned long long *d, h[8] = {0};
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    k_check<<<256, 256>>>(12345u, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mismatches: spread_codes %llu not_acgt %llu col_bytes %llu nibble_to_bytes %llu ok_bits4 %llu\n", h[0], h[1], h[2], h[3], h[4]);
    return (h[0] | h[1] | h[2] | h[3] | h[4]) ? 1 : 0;
}
