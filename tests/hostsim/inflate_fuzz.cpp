// inflate_fuzz -- TEST INFRASTRUCTURE: drives amplipy_amd/csrc/amp_inflate.hpp under AddressSanitizer / UBSan on the CPU
// (GPU sanitizers are not available on the pool): valid streams of every level / strategy must round-trip, mutated and random
// streams must be refused or give wrong bytes without any out-of-bounds access.  Input and output live in exactly-sized heap
// blocks so that the sanitizer sees the first byte outside them.
//   g++ -O1 -g -fsanitize=address,undefined -o inflate_fuzz inflate_fuzz.cpp -lz && ./inflate_fuzz [seconds]
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../amplipy_amd/csrc/amp_inflate.hpp"

static uint64_t st = 88172645463325252ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); }

int main(int argc, char **argv) {
    const double budget = argc > 1 ? atof(argv[1]) : 20.0;
    const auto t0 = std::chrono::steady_clock::now();
    ampinf::Tables tabs;
    long n_valid = 0, n_mut = 0, n_mut_ok = 0, n_rand = 0, bad = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget) {
        // a payload: runs, text-like bytes, noise, in random proportions
        const size_t n = rnd() % 8 == 0 ? rnd() % 70000 : rnd() % 3000;
        std::vector<uint8_t> data(n);
        const int kind = rnd() % 4;
        for (size_t i = 0; i < n; ++i)
            data[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? (uint8_t)(rnd() % 4) : kind == 2 ? (uint8_t)("ACGT"[rnd() % 4]) : (uint8_t)((i / 7) % 251);
        if (kind == 2 && n > 600) for (size_t i = 300; i < n; ++i) if (rnd() % 3) data[i] = data[i - 1 - rnd() % 299];
        const int level = (int)(rnd() % 10), strat = (int)(rnd() % 5);
        z_stream zs; memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, level, Z_DEFLATED, -15, 1 + (int)(rnd() % 9), strat);
        std::vector<uint8_t> comp(deflateBound(&zs, (uLong)n) + 64);
        zs.next_in = data.data(); zs.avail_in = (uInt)(n / 2); zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        deflate(&zs, rnd() % 2 ? Z_FULL_FLUSH : Z_NO_FLUSH);
        zs.avail_in = (uInt)(n - n / 2);
        deflate(&zs, Z_FINISH);
        const size_t cn = zs.total_out;
        deflateEnd(&zs);
        uint8_t *in = (uint8_t *)malloc(cn ? cn : 1), *out = (uint8_t *)malloc(n ? n : 1);
        memcpy(in, comp.data(), cn);
        ++n_valid;
        if (!ampinf::inflate_block(in, cn, out, n, tabs) || (n && memcmp(out, data.data(), n) != 0)) { ++bad; fprintf(stderr, "round trip failed: n %zu level %d strategy %d\n", n, level, strat); }
        for (int m = 0; m < 6 && cn; ++m) {                 // mutations of the valid stream, sometimes with a wrong expected size
            uint8_t *mi = (uint8_t *)malloc(cn);
            memcpy(mi, in, cn);
            const int flips = 1 + (int)(rnd() % 3);
            for (int f = 0; f < flips; ++f) mi[rnd() % cn] ^= (uint8_t)(1u << (rnd() % 8));
            const size_t cut = rnd() % 4 == 0 ? rnd() % cn + 1 : cn;
            const size_t want = rnd() % 4 == 0 ? rnd() % (n + 100) : n;
            uint8_t *mo = (uint8_t *)malloc(want ? want : 1);
            uint8_t *mc = (uint8_t *)malloc(cut);
            memcpy(mc, mi, cut);
            ++n_mut;
            if (ampinf::inflate_block(mc, cut, mo, want, tabs)) ++n_mut_ok;
            free(mi); free(mo); free(mc);
        }
        {                                                   // random bytes
            const size_t rn = 1 + rnd() % 2000, want = rnd() % 70000;
            uint8_t *ri = (uint8_t *)malloc(rn), *ro = (uint8_t *)malloc(want ? want : 1);
            for (size_t i = 0; i < rn; ++i) ri[i] = (uint8_t)rnd();
            if (rnd() % 2) ri[0] = (uint8_t)((ri[0] & ~7u) | (rnd() % 2) | 4u);      // often a dynamic-Huffman header
            ++n_rand;
            (void)ampinf::inflate_block(ri, rn, ro, want, tabs);
            free(ri); free(ro);
        }
        free(in); free(out);
    }
    printf("valid streams %ld (failed %ld), mutated %ld (still accepted %ld), random %ld\n", n_valid, bad, n_mut, n_mut_ok, n_rand);
    return bad ? 1 : 0;
}
