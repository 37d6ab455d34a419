// ampbam.cpp -- BAM/BGZF decode into the packed read batch and re-encode of trimmed records.
// See include/ampbam.h for what this replaces in the reference (pysam, AmpliPy.py:296-356, :896-911).
// Host only: zlib + std::thread.  The whole file is inflated into one buffer (an amplicon BAM at
// 100k x is a few GB; the GPU box has > 250 GB of RAM), so records are addressable by number and
// the writer can copy the unchanged parts of a record straight from the input image.
#include "../../include/ampbam.h"
#include "amp_inflate.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
inline void put16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

int pick_threads(int n) {
    if (n > 0) return std::min(n, 64);
    unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
}

template <class F>
void parallel_for(int n_threads, int64_t n_items, F &&fn) {   // fn(item) ; dynamic hand-out in chunks
    if (n_items <= 0) return;
    const int nt = (int)std::min<int64_t>(n_threads, n_items);
    if (nt <= 1) { for (int64_t i = 0; i < n_items; ++i) fn(i); return; }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&]() { for (;;) { int64_t i = next.fetch_add(1); if (i >= n_items) break; fn(i); } });
    for (auto &x : th) x.join();
}

// SAMv1 5.3: bin of the 0-based half-open interval [beg, end)
inline uint16_t reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// DEFLATE through libdeflate when the shared object is present (2-3x zlib on 64 KiB blocks; htslib
// makes the same choice), resolved at run time so that there is no build-time dependency; zlib otherwise.
// AMPBAM_ZLIB=1 in the environment forces zlib.
struct LibDeflate {
    void *(*alloc_d)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_d)(void *) = nullptr;
    void *(*alloc_c)(int) = nullptr;
    size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*free_c)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        const char *force = std::getenv("AMPBAM_ZLIB");
        if (force && force[0] && force[0] != '0') return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc_d = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_d = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        alloc_c = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        free_c = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc_d && decompress && free_d && alloc_c && compress && free_c && crc;
    }
};
const LibDeflate &libdeflate() { static const LibDeflate L; return L; }

// CRC-32 (the gzip polynomial) of a block.  zlib 1.2.11's table-driven crc32 runs at 1 GB/s, its inflate at 0.4 GB/s: the check
// was a quarter of the inflate stage.  On x86 with PCLMULQDQ the bulk of a block is folded 64 bytes at a time with
// carry-less multiplications (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel
// 2009: fold constants x^(n) mod P for n = 4*128+64, 4*128, 128+64, 128, 96, 64 bits, then a Barrett reduction; the constants
// below are those of the bit-reflected polynomial 0xEDB88320), the tail of less than 16 bytes and short inputs by zlib;
// anything else uses zlib throughout.  tests/test_bam_native.py compares the two on random lengths and contents.
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_fold_pclmul(const uint8_t *buf, size_t len, uint32_t state) {
    // len >= 64 and a multiple of 16; state = the running register (the complement of the CRC so far)
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                  // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);               // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
bool have_pclmul() {
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !(std::getenv("AMPBAM_ZLIB_CRC") && std::getenv("AMPBAM_ZLIB_CRC")[0] == '1');
    return ok;
}
#else
bool have_pclmul() { return false; }
uint32_t crc32_fold_pclmul(const uint8_t *, size_t, uint32_t s) { return s; }
#endif
uint32_t crc32_block(const uint8_t *p, size_t n) {
    if (n < 64 || !have_pclmul()) return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
    const size_t bulk = n & ~(size_t)15;
    const uint32_t c = ~crc32_fold_pclmul(p, bulk, 0xFFFFFFFFu);
    return bulk == n ? c : (uint32_t)crc32(c, p + bulk, (uInt)(n - bulk));
}

// one worker's inflater: the block decoder of amp_inflate.hpp first; a block it refuses, or whose CRC then differs, goes
// through zlib (libdeflate when present) as before.  AMPBAM_ZLIB_INFLATE=1 keeps the library decoder for every block.
bool use_own_inflate() {
    static const bool on = !(std::getenv("AMPBAM_ZLIB_INFLATE") && std::getenv("AMPBAM_ZLIB_INFLATE")[0] == '1');
    return on;
}
struct Inflater {
    void *ld = nullptr;
    z_stream zs;
    bool z_ok = false;
    ampinf::Tables tabs;
    Inflater() {
        if (libdeflate().ok) ld = libdeflate().alloc_d();
        if (!ld) { std::memset(&zs, 0, sizeof(zs)); z_ok = inflateInit2(&zs, -15) == Z_OK; }
    }
    ~Inflater() { if (ld) libdeflate().free_d(ld); else if (z_ok) inflateEnd(&zs); }
    bool run(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, uint32_t want_crc) {
        if (use_own_inflate() && ampinf::inflate_block(in, in_len, out, out_len, tabs) && crc32_block(out, out_len) == want_crc) return true;
        if (ld) {
            size_t got = 0;
            if (libdeflate().decompress(ld, in, in_len, out, out_len, &got) != 0 || got != out_len) return false;
            return crc32_block(out, out_len) == want_crc;
        }
        if (!z_ok) return false;
        inflateReset(&zs);
        zs.next_in = const_cast<Bytef *>(in); zs.avail_in = (uInt)in_len;
        zs.next_out = out; zs.avail_out = (uInt)out_len;
        if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) return false;
        return crc32_block(out, out_len) == want_crc;
    }
};

const uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct Block { size_t in_off, in_len, out_off, out_len; uint32_t crc; };

// Big buffers are kept for the next taker instead of being returned to the system.  Releasing one is a munmap, and when the
// HIP runtime has copied from it (the packed batch goes to the GPU from these very pages) the GPU driver must first tear
// down its own mapping of them: measured on the GPU box, the HIP call that followed the close of a 1.5 M-read file waited
// 30-65 ms for that, a third of a `variants` run -- whichever call it was (hipHostMalloc, a kernel launch, hipFree; a
// 100 ms sleep in between made it vanish, and so did not closing the file).  A piece's buffers also serve the next piece
// already faulted in.  AMPBAM_POOL_MB bounds what is held (default 4096).
struct BytePool {
    std::mutex mu;
    std::vector<std::pair<uint8_t *, size_t>> idle;
    size_t held = 0, limit;
    BytePool() { const char *e = std::getenv("AMPBAM_POOL_MB"); limit = (size_t)(e && e[0] ? std::strtoull(e, nullptr, 10) : 4096ull) << 20; }
    ~BytePool() { for (auto &b : idle) std::free(b.first); }
    uint8_t *take(size_t want, size_t &cap) {
        std::lock_guard<std::mutex> g(mu);
        int best = -1;
        for (int i = 0; i < (int)idle.size(); ++i)
            if (idle[(size_t)i].second >= want && (best < 0 || idle[(size_t)i].second < idle[(size_t)best].second)) best = i;
        if (best < 0) return nullptr;
        uint8_t *p = idle[(size_t)best].first; cap = idle[(size_t)best].second;
        held -= cap;
        idle.erase(idle.begin() + best);
        return p;
    }
    void give(uint8_t *p, size_t cap) {
        {
            std::lock_guard<std::mutex> g(mu);
            if (held + cap <= limit) { idle.emplace_back(p, cap); held += cap; return; }
        }
        std::free(p);
    }
};
BytePool &byte_pool() { static BytePool P; return P; }

// byte buffer that is NOT zero-filled on allocation (the inflated image is tens of MB to GB).  Buffers of 4 MB and more are
// 2 MB-aligned and marked MADV_HUGEPAGE: sixteen threads filling a fresh 400 MB image take 100,000 page faults of 4 KB under
// one mmap lock; with 2 MB pages they are 500 times fewer.  They come from and go back to the pool above.
struct Bytes {
    static constexpr size_t HUGE = (size_t)2 << 20;
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    Bytes() = default;
    Bytes(const Bytes &) = delete;
    Bytes &operator=(const Bytes &) = delete;
    ~Bytes() { drop(); }
    void drop() {
        if (p && cap >= 2 * HUGE) byte_pool().give(p, cap); else std::free(p);
        p = nullptr; n = cap = 0;
    }
    bool resize(size_t m) {
        if (m > cap) {
            if (m >= 2 * HUGE) {
                size_t want = (m + m / 8 + HUGE - 1) & ~(HUGE - 1);
                uint8_t *q = byte_pool().take(m, want);
                if (!q) {
                    q = (uint8_t *)std::aligned_alloc(HUGE, want);
                    if (!q) return false;
                    if (std::getenv("AMPBAM_HUGEPAGES")) (void)madvise(q, want, MADV_HUGEPAGE);
                }
                if (n) memcpy(q, p, n);
                const size_t keep = n;
                drop();
                p = q; cap = want; n = keep;
            } else {
                uint8_t *q = (uint8_t *)std::realloc(p, m);
                if (!q) return false;
                p = q; cap = m;
            }
        }
        n = m;
        return true;
    }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

// typed array on a Bytes buffer: not zero-filled, pooled like it
template <class T>
struct Arr {
    Bytes b;
    size_t n = 0;
    bool resize(size_t m) { if (!b.resize(m * sizeof(T))) return false; n = m; return true; }
    T *data() { return (T *)b.data(); }
    const T *data() const { return (const T *)b.data(); }
    size_t size() const { return n; }
    T &operator[](size_t i) { return ((T *)b.data())[i]; }
    const T &operator[](size_t i) const { return ((const T *)b.data())[i]; }
};

}  // namespace

struct ampbam_file {
    Bytes data;                             // the inflated stream
    size_t text_off = 0, text_len = 0;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lens;
    std::vector<uint64_t> rec_off;          // offset of every record's block_size field (+ end sentinel)
    std::vector<uint64_t> rec_info;         // l_seq | n_cigar_op << 32 | flag << 48, so that planning a batch does not touch the image
    int n_threads = 1;
    std::string err;
    // ampbam_open_range: the part's place in the whole inflated stream
    uint64_t img_base = 0;                  // inflated offset of data[0]
    uint64_t part_first = 0, part_end = 0;  // inflated offsets: this part's first record; one past its last record
    Bytes hdr;                              // header bytes of parts that do not start at the file's first block
    const uint8_t *text_ptr = nullptr;
    // decode outputs (reused)
    Arr<int32_t> pos, tlen;
    Arr<uint16_t> flag;
    Arr<uint32_t> lseq, cig;
    Arr<uint64_t> cig_off, seq_off;
    Bytes seq, qual;
    Arr<int64_t> src_index;
};

struct ampbam_writer {
    FILE *fp = nullptr;
    int level = -1, n_threads = 1;
    Bytes pend;                             // uncompressed bytes not yet written (no zero fill on growth)
    Bytes comp;                             // compressed blocks of one flush, at a fixed stride
    int64_t header_bytes = 0;               // compressed bytes of the header's own blocks
    std::string err;
};

extern "C" {

int ampbam_version(void) { return 1; }
int ampbam_inflate_raw(const void *in, int64_t n_in, void *out, int64_t n_out) {
    if (!in || !out || n_in < 0 || n_out < 0) return AMPBAM_EINVAL;
    static thread_local ampinf::Tables tabs;
    return ampinf::inflate_block((const uint8_t *)in, (size_t)n_in, (uint8_t *)out, (size_t)n_out, tabs) ? AMPBAM_OK : AMPBAM_EFORMAT;
}
uint32_t ampbam_crc32(const void *data, int64_t n_bytes) { return (data && n_bytes > 0) ? crc32_block((const uint8_t *)data, (size_t)n_bytes) : 0u; }

const char *ampbam_strerror(int rc) {
    switch (rc) {
        case AMPBAM_OK: return "ok";
        case AMPBAM_EINVAL: return "invalid argument";
        case AMPBAM_EIO: return "I/O error";
        case AMPBAM_EFORMAT: return "not a valid BGZF/BAM stream";
        case AMPBAM_ENOMEM: return "out of memory";
        default: return "unknown error";
    }
}

const char *ampbam_last_error(const ampbam_file *f) { return f ? f->err.c_str() : ""; }

void ampbam_close(ampbam_file *f) {
    if (!f) return;
    // With the buffer pool (the default) closing only hands the buffers back.  Without it (AMPBAM_POOL_MB=0) returning hundreds
    // of MB of touched pages to the kernel takes ~0.3 ms per MB, which round 2 kept off the caller's clock with a thread.
    if (byte_pool().limit == 0 && f->data.cap > ((size_t)64 << 20)) std::thread([f]() { delete f; }).detach();
    else delete f;
}

int ampbam_open(const char *path, int n_threads, ampbam_file **out) { return ampbam_open_range(path, n_threads, 0, 1, out); }

int64_t ampbam_n_records(const ampbam_file *f) { return f ? (int64_t)f->rec_off.size() - 1 : 0; }

int ampbam_header_text(const ampbam_file *f, const char **text, int64_t *len) {
    if (!f || !text || !len) return AMPBAM_EINVAL;
    *text = f->text_ptr ? (const char *)f->text_ptr : (const char *)f->data.data() + f->text_off;
    size_t n = f->text_len;
    while (n && (*text)[n - 1] == '\0') --n;      // some writers NUL-pad the text
    *len = (int64_t)n;
    return AMPBAM_OK;
}

int32_t ampbam_n_refs(const ampbam_file *f) { return f ? (int32_t)f->ref_names.size() : 0; }

int ampbam_ref(const ampbam_file *f, int32_t i, const char **name, int32_t *length) {
    if (!f || i < 0 || i >= (int32_t)f->ref_names.size()) return AMPBAM_EINVAL;
    if (name) *name = f->ref_names[(size_t)i].c_str();
    if (length) *length = f->ref_lens[(size_t)i];
    return AMPBAM_OK;
}

int ampbam_decode(ampbam_file *f, int64_t first, int64_t count, ampbam_batch *out) {
    if (!f || !out || first < 0 || count < 0 || first + count > ampbam_n_records(f)) return AMPBAM_EINVAL;
    const uint8_t *d = f->data.data();
    // pass 1 (fixed fields only, from the index): which records are rows, and where their variable parts go.  Per chunk of
    // 4096 records on the threads, an exclusive scan over the chunks, then every chunk fills its own rows in pass 2 (a serial
    // loop with three push_backs per record was a third of the decode of 1.5 M records).
    const int64_t grain = 4096, n_chunks = (count + grain - 1) / grain;
    struct Plan { uint64_t rows, cig, bases; };
    std::vector<Plan> plan;
    try { plan.assign((size_t)n_chunks + 1, Plan{0, 0, 0}); } catch (const std::bad_alloc &) { return AMPBAM_ENOMEM; }
    parallel_for(f->n_threads, n_chunks, [&](int64_t ch) {
        Plan p{0, 0, 0};
        for (int64_t r = first + ch * grain; r < std::min(first + count, first + (ch + 1) * grain); ++r) {
            const uint64_t info = f->rec_info[(size_t)r];
            const uint32_t l_seq = (uint32_t)info, n_cig = (uint32_t)(info >> 32) & 0xFFFFu, flag = (uint32_t)(info >> 48);
            if ((flag & 4u) || n_cig == 0) continue;                                    // AmpliPy.py:902
            ++p.rows; p.cig += n_cig; p.bases += ((uint64_t)l_seq + 7) & ~7ull;
        }
        plan[(size_t)ch] = p;
    });
    {
        Plan run{0, 0, 0};
        for (int64_t ch = 0; ch <= n_chunks; ++ch) { const Plan p = plan[(size_t)ch]; plan[(size_t)ch] = run; run.rows += p.rows; run.cig += p.cig; run.bases += p.bases; }
    }
    const int64_t n = (int64_t)plan[(size_t)n_chunks].rows;
    const uint64_t co = plan[(size_t)n_chunks].cig, so = plan[(size_t)n_chunks].bases;
    if (!f->src_index.resize((size_t)n) || !f->cig_off.resize((size_t)n + 1) || !f->seq_off.resize((size_t)n + 1) || !f->pos.resize((size_t)n) ||
        !f->tlen.resize((size_t)n) || !f->flag.resize((size_t)n) || !f->lseq.resize((size_t)n) || !f->cig.resize((size_t)co + 4) ||
        !f->seq.resize((size_t)(so / 2) + 16) || !f->qual.resize((size_t)so + 16)) return AMPBAM_ENOMEM;
    f->cig_off[0] = 0; f->seq_off[0] = 0;
    std::memset(f->cig.data() + co, 0, 16);
    std::memset(f->seq.data() + so / 2, 0, 16); std::memset(f->qual.data() + so, 0, 16);
    // pass 2 (parallel): copy
    parallel_for(f->n_threads, n_chunks, [&](int64_t ch) {
        int64_t i = (int64_t)plan[(size_t)ch].rows;
        uint64_t c_at = plan[(size_t)ch].cig, s_at = plan[(size_t)ch].bases;
        for (int64_t r = first + ch * grain; r < std::min(first + count, first + (ch + 1) * grain); ++r) {
            const uint64_t info = f->rec_info[(size_t)r];
            if (((uint32_t)(info >> 48) & 4u) || ((uint32_t)(info >> 32) & 0xFFFFu) == 0) continue;
            const uint8_t *c = d + f->rec_off[(size_t)r] + 4;
            const uint32_t l_name = c[8], n_cig = le16(c + 12), l_seq = le32(c + 16);
            const uint64_t padded = ((uint64_t)l_seq + 7) & ~7ull;                        // bases, multiple of 8
            f->src_index[(size_t)i] = r;
            f->cig_off[(size_t)i + 1] = c_at + n_cig; f->seq_off[(size_t)i + 1] = s_at + padded;
            f->pos[(size_t)i] = (int32_t)le32(c + 4);
            f->flag[(size_t)i] = le16(c + 14);
            f->lseq[(size_t)i] = l_seq;
            f->tlen[(size_t)i] = (int32_t)le32(c + 28);
            const uint8_t *v = c + 32 + l_name;
            std::memcpy(&f->cig[(size_t)c_at], v, 4ull * n_cig);                          // little-endian host
            v += 4ull * n_cig;
            uint8_t *sq = f->seq.data() + s_at / 2, *ql = f->qual.data() + s_at;
            std::memcpy(sq, v, (l_seq + 1) / 2);
            if (l_seq & 1) sq[l_seq / 2] &= 0xF0;                                      // spare nibble and padding are zero in the batch
            std::memset(sq + (l_seq + 1) / 2, 0, (size_t)(padded / 2 - (l_seq + 1) / 2));
            v += (l_seq + 1) / 2;
            std::memcpy(ql, v, l_seq);
            std::memset(ql + l_seq, 0, (size_t)(padded - l_seq));
            ++i; c_at += n_cig; s_at += padded;
        }
    });
    out->n_reads = n;
    out->pos = f->pos.data(); out->flag = f->flag.data(); out->tlen = f->tlen.data(); out->lseq = f->lseq.data();
    out->cig_off = f->cig_off.data(); out->cig = f->cig.data(); out->seq_off = f->seq_off.data();
    out->seq = f->seq.data(); out->qual = f->qual.data(); out->src_index = f->src_index.data();
    out->n_cig = (int64_t)co; out->n_bases = (int64_t)so;
    out->n_skipped = count - n;
    return AMPBAM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// one part of a BAM file (range partition of a run over ranks, or a file read piece by piece)
// ---------------------------------------------------------------------------------------------
namespace {

struct MappedFile {
    const uint8_t *p = nullptr; size_t n = 0;
    ~MappedFile() { if (p && n) munmap(const_cast<uint8_t *>(p), n); }
    int map(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return AMPBAM_EIO;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 0) { ::close(fd); return AMPBAM_EIO; }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); n = 0; return AMPBAM_EIO; }
            p = (const uint8_t *)m;
        }
        ::close(fd);
        return AMPBAM_OK;
    }
};

// the BGZF block table of a mapped file (a serial hop over the block headers; nothing is inflated)
int block_table(const MappedFile &raw, std::vector<Block> &blocks, size_t &total) {
    size_t p = 0;
    total = 0;
    while (p < raw.n) {
        if (raw.n - p < 18 || raw.p[p] != 0x1f || raw.p[p + 1] != 0x8b || raw.p[p + 2] != 8 || !(raw.p[p + 3] & 4)) return AMPBAM_EFORMAT;
        const size_t xlen = le16(&raw.p[p + 10]);
        if (raw.n - p < 12 + xlen) return AMPBAM_EFORMAT;
        size_t bsize = 0, q = p + 12;
        const size_t xend = p + 12 + xlen;
        while (q + 4 <= xend) {
            const size_t slen = le16(&raw.p[q + 2]);
            if (raw.p[q] == 'B' && raw.p[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = (size_t)le16(&raw.p[q + 4]) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || raw.n - p < bsize) return AMPBAM_EFORMAT;
        Block b;
        b.in_off = p + 12 + xlen; b.in_len = bsize - 12 - xlen - 8;
        b.crc = le32(&raw.p[p + bsize - 8]); b.out_len = le32(&raw.p[p + bsize - 4]);
        b.out_off = total;
        if (b.out_len > 65536) return AMPBAM_EFORMAT;
        total += b.out_len;
        blocks.push_back(b);
        p += bsize;
    }
    return AMPBAM_OK;
}

// Does a plausible BAM record start at d[o]?  (SAMv1 4.2: block_size, refID, pos, l_read_name, mapq, bin, n_cigar_op, flag,
// l_seq, next_refID, next_pos, tlen, read_name NUL-terminated, CIGAR ops 0..8.)  *next = offset behind it.
bool plausible_record(const uint8_t *d, size_t avail, size_t o, int32_t n_ref, size_t *next) {
    if (o + 36 > avail) return false;
    const size_t bs = le32(d + o);
    if (bs < 32 || bs > (1u << 27)) return false;
    const uint8_t *c = d + o + 4;
    const int32_t ref_id = (int32_t)le32(c), pos = (int32_t)le32(c + 4), next_ref = (int32_t)le32(c + 20), next_pos = (int32_t)le32(c + 24);
    const uint32_t l_name = c[8], n_cig = le16(c + 12), l_seq = le32(c + 16);
    if (ref_id < -1 || ref_id >= n_ref || next_ref < -1 || next_ref >= n_ref || pos < -1 || next_pos < -1 || l_name < 1) return false;
    if (32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq > bs) return false;
    if (o + 4 + 32 + l_name + 4ull * n_cig > avail) { *next = o + 4 + bs; return true; }      // the fixed part is all we can see
    if (c[32 + l_name - 1] != 0) return false;
    for (uint32_t k = 0; k + 1 < l_name; ++k) if (c[32 + k] < 33 || c[32 + k] > 126) return false;
    for (uint32_t k = 0; k < n_cig; ++k) if ((le32(c + 32 + l_name + 4 * k) & 15u) > 8u) return false;
    *next = o + 4 + bs;
    return true;
}

}  // namespace

extern "C" {

// ampbam.h: part `part` of `n_parts` of the file.  The cut points are compressed-byte offsets rounded up to BGZF block
// starts (for a coordinate-sorted BAM of similar reads: equal shares of bases); a part owns the records that START inside
// its blocks.  Where the first record of a part starts is not written anywhere in a BAM file: it is found by looking for
// the first offset from which a chain of plausible records runs (what splitting BAM readers do), and the caller makes it
// exact by checking that every part's first record starts where the part before it ended (ampbam_part_range).
// Record index of image bytes [o0, lim) on several threads.  A BAM record only says where the NEXT one starts, so a thread that
// begins in the middle has to guess: the first offset from which 64 plausible records follow each other (the test the parts
// of a file use).  The guess is never trusted: the lists are stitched in order, a list is taken only if it starts exactly
// where the chain from the true first record has arrived, and where it does not the chain is walked serially until it meets
// the list (or passes it).  Records are taken while their fixed fields and their whole body lie inside [0, avail); the walk
// stops at the first one that does not (the caller's serial loop, which can inflate more, goes on from *o_end).  A serial walk
// of 1.5 M records behind sixteen inflating threads took as long as the inflating itself.
bool index_records(const uint8_t *d, size_t avail, size_t o0, size_t lim, int32_t n_ref, int n_threads,
                   std::vector<uint64_t> &rec_off, std::vector<uint64_t> &rec_info, size_t *o_end) {
    struct Seg { std::vector<uint64_t> off, info; size_t end = 0; bool stopped = false, bad = false; };
    auto step = [&](size_t o, uint64_t *info, size_t *next, bool *bad) -> bool {      // one record at o: false = stop here
        if (o + 36 > avail) return false;
        const size_t bs = le32(d + o);
        if (bs < 32 || bs > (1u << 27)) { *bad = true; return false; }      // (the bound of plausible_record: no record is 128 MB long)
        if (o + 4 + bs > avail) return false;
        const uint8_t *c = d + o + 4;
        const uint64_t n_cig = le16(c + 12), flag = le16(c + 14), l_seq = le32(c + 16), l_name = c[8];
        if (32ull + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq > bs) { *bad = true; return false; }
        *info = l_seq | (n_cig << 32) | (flag << 48);
        *next = o + 4 + bs;
        return true;
    };
    const int T = (lim > o0 && lim - o0 >= ((size_t)4 << 20)) ? std::max(1, std::min(n_threads, 64)) : 1;
    std::vector<Seg> seg((size_t)T);
    auto bound = [&](int t) { return t >= T ? lim : o0 + (size_t)((unsigned __int128)(lim - o0) * (unsigned)t / (unsigned)T); };
    parallel_for(T, T, [&](int64_t t) {
        Seg &g = seg[(size_t)t];
        const size_t hi = bound((int)t + 1);
        size_t o = o0;
        if (t > 0) {
            bool found = false;
            for (size_t cand = bound((int)t); cand < hi && !found; ++cand) {
                size_t at = cand, nx = 0;
                int chain = 0;
                while (chain < 64 && at + 36 <= avail && plausible_record(d, avail, at, n_ref, &nx)) { at = nx; ++chain; }
                if (chain >= 64 || (chain >= 1 && at + 36 > avail)) { o = cand; found = true; }
            }
            if (!found) { g.end = hi; g.stopped = true; return; }        // (stitching walks this stretch serially)
        }
        if (hi > o) { g.off.reserve((hi - o) / 200 + 16); g.info.reserve((hi - o) / 200 + 16); }
        while (o < hi) {
            uint64_t info; size_t next;
            if (!step(o, &info, &next, &g.bad)) { g.stopped = true; break; }
            g.off.push_back(o); g.info.push_back(info);
            o = next;
        }
        g.end = o;
    });
    size_t cur = o0;
    bool more = true;
    for (int t = 0; t < T && more; ++t) {
        Seg &g = seg[(size_t)t];
        size_t p = 0;
        // the chain has to meet the list: walk it until it does (at once when the guess was right)
        for (;;) {
            while (p < g.off.size() && g.off[p] < cur) ++p;
            if (p < g.off.size() && g.off[p] == cur) break;
            if (cur >= bound(t + 1) && p >= g.off.size() && !g.stopped) break;       // the list is used up and the chain is past the segment
            if (cur >= lim) { more = false; break; }
            if (cur >= bound(t + 1)) break;                                        // the chain left the segment without meeting the list
            uint64_t info; size_t next; bool bad = false;
            if (!step(cur, &info, &next, &bad)) { if (bad) return false; more = false; break; }
            rec_off.push_back(cur); rec_info.push_back(info);
            cur = next;
        }
        if (!more) break;
        if (p < g.off.size() && g.off[p] == cur) {
            rec_off.insert(rec_off.end(), g.off.begin() + (long)p, g.off.end());
            rec_info.insert(rec_info.end(), g.info.begin() + (long)p, g.info.end());
            cur = g.end;
            if (g.stopped) { if (g.bad) return false; more = false; }
        }
    }
    *o_end = cur;
    return true;
}

int ampbam_open_range(const char *path, int n_threads, int part, int n_parts, ampbam_file **out) {
    return ampbam_open_range_at(path, n_threads, part, n_parts, UINT64_MAX, out);
}

int ampbam_open_range_at(const char *path, int n_threads, int part, int n_parts, uint64_t first_hint, ampbam_file **out) {
    if (!path || !out || n_parts < 1 || part < 0 || part >= n_parts) return AMPBAM_EINVAL;
    *out = nullptr;
    MappedFile raw;
    int rc = raw.map(path);
    if (rc) return rc;
    // The BGZF block table is the same for every piece of a file: a process that walks a file piece by piece hops over the
    // block headers of the whole file ONCE (keyed by path, size and modification time), not once per piece -- on a 10 GB file
    // that was two minutes of page faults over 2,500 pieces
    std::vector<Block> blocks;
    size_t total = 0;
    {
        static std::mutex mu;
        static std::string c_path; static size_t c_size = 0; static int64_t c_mtime = 0; static size_t c_total = 0;
        static std::vector<Block> c_blocks;
        struct stat st;
        const bool have_st = ::stat(path, &st) == 0;
        const int64_t mt = have_st ? (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec : -1;
        std::lock_guard<std::mutex> lk(mu);
        if (have_st && c_path == path && c_size == raw.n && c_mtime == mt && !c_blocks.empty()) {
            blocks = c_blocks; total = c_total;
        } else {
            rc = block_table(raw, blocks, total);
            if (rc) return rc;
            if (have_st && n_parts > 1) { c_path = path; c_size = raw.n; c_mtime = mt; c_blocks = blocks; c_total = total; }
        }
    }
    const int64_t nb = (int64_t)blocks.size();
    if (nb == 0) return AMPBAM_EFORMAT;
    ampbam_file *f = new (std::nothrow) ampbam_file();
    if (!f) return AMPBAM_ENOMEM;
    f->n_threads = pick_threads(n_threads);
    auto fail = [&](int r) { delete f; return r; };
    auto cut = [&](int k) -> int64_t {                    // first block at or behind share k of the compressed bytes
        if (k <= 0) return 0;
        if (k >= n_parts) return nb;
        const size_t want = (size_t)((unsigned __int128)raw.n * (unsigned)k / (unsigned)n_parts);
        int64_t lo = 0, hi = nb;
        while (lo < hi) { const int64_t m = (lo + hi) / 2; if (blocks[(size_t)m].in_off >= want) hi = m; else lo = m + 1; }
        return lo;
    };
    const int64_t b_lo = cut(part), b_hi = cut(part + 1);
    // ---- the header: from the file's first blocks (a few KB), whatever the part -------------------------------------
    int32_t n_ref = 0;
    size_t hdr_end = 0;                                  // inflated offset behind the reference dictionary
    {
        Inflater inf;
        size_t have = 0;
        int64_t k = 0;
        auto more = [&]() -> bool {
            if (k >= nb) return false;
            const Block &b = blocks[(size_t)k++];
            if (!f->hdr.resize(have + b.out_len + 16)) return false;
            if (b.out_len && !inf.run(raw.p + b.in_off, b.in_len, f->hdr.data() + have, b.out_len, b.crc)) return false;
            have += b.out_len;
            return true;
        };
        auto need = [&](size_t upto) { while (have < upto) if (!more()) return false; return true; };
        if (!need(12) || std::memcmp(f->hdr.data(), "BAM\1", 4) != 0) return fail(AMPBAM_EFORMAT);
        const size_t l_text = le32(f->hdr.data() + 4);
        if (!need(12 + l_text)) return fail(AMPBAM_EFORMAT);
        size_t o = 8 + l_text;
        n_ref = (int32_t)le32(f->hdr.data() + o); o += 4;
        if (n_ref < 0) return fail(AMPBAM_EFORMAT);
        for (int32_t r = 0; r < n_ref; ++r) {
            if (!need(o + 4)) return fail(AMPBAM_EFORMAT);
            const size_t l_name = le32(f->hdr.data() + o); o += 4;
            if (l_name == 0 || !need(o + l_name + 4)) return fail(AMPBAM_EFORMAT);
            f->ref_names.emplace_back((const char *)(f->hdr.data() + o), l_name - 1); o += l_name;
            f->ref_lens.push_back((int32_t)le32(f->hdr.data() + o)); o += 4;
        }
        f->text_off = 8; f->text_len = l_text;
        f->text_ptr = f->hdr.data() + 8;
        hdr_end = o;
    }
    // ---- inflate the part's blocks, and behind them as many as its last record needs -----------------------------------
    const uint64_t end_off = b_hi < nb ? blocks[(size_t)b_hi].out_off : total;      // records that start below this are the part's
    int64_t b_ext = std::min<int64_t>(nb, b_hi + 2);
    f->img_base = b_lo < nb ? blocks[(size_t)b_lo].out_off : total;
    f->part_first = f->part_end = f->img_base;
    int64_t inflated_to = b_lo;
    auto inflate_to = [&](int64_t upto) -> int {          // data covers blocks [b_lo, upto)
        upto = std::min<int64_t>(upto, nb);
        if (upto <= inflated_to) return AMPBAM_OK;
        const size_t bytes = (size_t)((upto < nb ? blocks[(size_t)upto].out_off : total) - f->img_base);
        if (!f->data.resize(bytes + 16)) return AMPBAM_ENOMEM;
        std::memset(f->data.data() + bytes, 0, 16);
        std::atomic<int> bad{0};
        const int64_t first = inflated_to, cnt = upto - inflated_to, grain = 8;
        parallel_for(f->n_threads, (cnt + grain - 1) / grain, [&](int64_t ch) {
            Inflater local;
            for (int64_t k = first + ch * grain; k < std::min(upto, first + (ch + 1) * grain); ++k) {
                const Block &b = blocks[(size_t)k];
                if (b.out_len && !local.run(raw.p + b.in_off, b.in_len, f->data.data() + (b.out_off - f->img_base), b.out_len, b.crc)) bad = 1;
            }
        });
        if (bad) return AMPBAM_EFORMAT;
        inflated_to = upto;
        return AMPBAM_OK;
    };
    if (b_lo >= nb || b_lo >= b_hi) {                     // an empty part (more parts than blocks)
        f->rec_off.push_back(0);
        *out = f;
        return AMPBAM_OK;
    }
    rc = inflate_to(b_ext);
    if (rc) return fail(rc);
    // ---- the part's first record, and the index of the records that start inside the part --------------------------------------
    size_t o = 0;                                         // offset in data
    const size_t lim = (size_t)(end_off - f->img_base);
    try {
        f->rec_off.reserve(lim / 200 + 16); f->rec_info.reserve(lim / 200 + 16);
        if (first_hint != UINT64_MAX && part > 0) {
            // the caller knows where this part's first record starts (the part before it ended there): no guessing
            if (first_hint < f->img_base) return fail(AMPBAM_EINVAL);
            if (first_hint >= end_off) {                   // no record starts in this part
                f->part_first = f->part_end = first_hint;
                f->rec_off.push_back(0);
                *out = f;
                return AMPBAM_OK;
            }
            o = (size_t)(first_hint - f->img_base);
            f->part_first = first_hint;
            if (!index_records(f->data.data(), f->data.size() - 16, o, lim, n_ref, f->n_threads, f->rec_off, f->rec_info, &o)) return fail(AMPBAM_EFORMAT);
        } else if (part == 0) {
            if (hdr_end < f->img_base) return fail(AMPBAM_EFORMAT);
            if (hdr_end >= end_off) {                      // the header (text + reference dictionary) fills the part's blocks and more: no record starts here
                f->part_first = f->part_end = hdr_end;
                f->rec_off.push_back(0);
                *out = f;
                return AMPBAM_OK;
            }
            o = hdr_end - (size_t)f->img_base;
            f->part_first = f->img_base + o;
            if (!index_records(f->data.data(), f->data.size() - 16, o, lim, n_ref, f->n_threads, f->rec_off, f->rec_info, &o)) return fail(AMPBAM_EFORMAT);
        } else {
            // a record that starts in front of this part may end anywhere in its first blocks: the first offset from which 64
            // plausible records follow each other (or run to the end of what is inflated) AND from which the whole part can be
            // indexed -- bytes inside a record can look like a run of records (a decoy, or aligned binary tags), but a chain
            // that starts on them runs into garbage before the part ends
            const size_t avail = f->data.size() - 16;
            const size_t limit = std::min<size_t>(avail, lim);
            bool found = false;
            const size_t cand0 = hdr_end > f->img_base ? (size_t)(hdr_end - f->img_base) : 0;      // (a header that reaches into this part)
            for (size_t cand = cand0; cand < limit && !found; ++cand) {
                size_t at = cand, nx = 0;
                int chain = 0;
                while (chain < 64 && at + 36 <= avail && plausible_record(f->data.data(), avail, at, n_ref, &nx)) { at = nx; ++chain; }
                if (!(chain >= 64 || (chain >= 1 && at + 36 > avail))) continue;
                f->rec_off.clear(); f->rec_info.clear();
                size_t oe = cand;
                if (!index_records(f->data.data(), avail, cand, lim, n_ref, f->n_threads, f->rec_off, f->rec_info, &oe)) continue;
                f->part_first = f->img_base + cand;
                o = oe; found = true;
            }
            if (!found) {                                  // no record starts in this part (one huge record spans it)
                f->rec_off.clear(); f->rec_info.clear();
                f->part_first = f->part_end = end_off;
                f->rec_off.push_back(0);
                *out = f;
                return AMPBAM_OK;
            }
        }
        while (o < lim) {                                 // what is left: the records that need more of the file inflated
            if (o + 36 > f->data.size() - 16) { rc = inflate_to(inflated_to + 4); if (rc) return fail(rc); if (o + 36 > f->data.size() - 16) return fail(AMPBAM_EFORMAT); }
            const uint8_t *d = f->data.data();
            const size_t bs = le32(d + o);
            if (bs < 32 || bs > (1u << 27)) return fail(AMPBAM_EFORMAT);
            while (o + 4 + bs > f->data.size() - 16) {
                if (inflated_to >= nb) return fail(AMPBAM_EFORMAT);
                rc = inflate_to(inflated_to + std::max<int64_t>(4, (int64_t)(bs / 60000)));
                if (rc) return fail(rc);
            }
            d = f->data.data();
            const uint8_t *c = d + o + 4;
            const uint64_t n_cig = le16(c + 12), flag = le16(c + 14), l_seq = le32(c + 16), l_name = c[8];
            if (32ull + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq > bs) return fail(AMPBAM_EFORMAT);
            f->rec_off.push_back(o);
            f->rec_info.push_back(l_seq | (n_cig << 32) | (flag << 48));
            o += 4 + bs;
        }
    } catch (const std::bad_alloc &) { return fail(AMPBAM_ENOMEM); }
    f->rec_off.push_back(o);
    f->part_end = f->img_base + o;
    *out = f;
    return AMPBAM_OK;
}

int ampbam_part_range(const ampbam_file *f, uint64_t *first, uint64_t *end) {
    if (!f || !first || !end) return AMPBAM_EINVAL;
    *first = f->part_first; *end = f->part_end;
    return AMPBAM_OK;
}

}  // extern "C"

extern "C" {

// ---------------------------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------------------------
static bool append(Bytes &b, const void *src, size_t n) {
    const size_t at = b.size();
    if (at + n > b.cap) {
        size_t want = std::max(at + n, b.cap + b.cap / 2 + 4096);
        uint8_t *q = (uint8_t *)std::realloc(b.p, want);
        if (!q) return false;
        b.p = q; b.cap = want;
    }
    std::memcpy(b.p + at, src, n);
    b.n = at + n;
    return true;
}

static int flush_blocks(ampbam_writer *w, bool all) {
    const size_t BS = 0xFF00, STRIDE = 65536 + 64;     // a BGZF block is at most 64 KiB
    const size_t nfull = w->pend.size() / BS, nblk = nfull + ((all && w->pend.size() % BS) ? 1 : 0);
    if (nblk == 0) return AMPBAM_OK;
    if (!w->comp.resize(nblk * STRIDE)) return AMPBAM_ENOMEM;
    std::vector<uint32_t> out_len(nblk, 0);
    std::atomic<int> bad{0};
    const int64_t grain = 4;
    parallel_for(w->n_threads, ((int64_t)nblk + grain - 1) / grain, [&](int64_t ch) {
        void *lc = libdeflate().ok ? libdeflate().alloc_c(w->level < 0 ? 6 : (w->level == 0 ? 1 : w->level)) : nullptr;
        for (int64_t k = ch * grain; k < std::min<int64_t>((int64_t)nblk, (ch + 1) * grain); ++k) {
            const size_t off = (size_t)k * BS, len = std::min(BS, w->pend.size() - off);
            uint8_t *o = w->comp.data() + (size_t)k * STRIDE;
            const size_t room = STRIDE - 18 - 8;
            size_t clen = 0;
            if (lc) {
                clen = libdeflate().compress(lc, w->pend.data() + off, len, o + 18, room);
            } else {
                z_stream zs;
                std::memset(&zs, 0, sizeof(zs));
                if (deflateInit2(&zs, w->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; break; }
                zs.next_in = w->pend.data() + off; zs.avail_in = (uInt)len;
                zs.next_out = o + 18; zs.avail_out = (uInt)room;
                const int rc = deflate(&zs, Z_FINISH);
                clen = rc == Z_STREAM_END ? zs.total_out : 0;
                deflateEnd(&zs);
            }
            if (clen == 0 || clen + 26 > 65536) { bad = 1; break; }
            const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            std::memcpy(o, hdr, 16);
            put16(o + 16, (uint16_t)(clen + 25));
            const uint32_t crc = crc32_block(w->pend.data() + off, len);
            put32(o + 18 + clen, crc);
            put32(o + 18 + clen + 4, (uint32_t)len);
            out_len[(size_t)k] = (uint32_t)(18 + clen + 8);
        }
        if (lc) libdeflate().free_c(lc);
    });
    if (bad) return AMPBAM_EIO;
    for (size_t k = 0; k < nblk; ++k)
        if (std::fwrite(w->comp.data() + k * STRIDE, 1, out_len[k], w->fp) != out_len[k]) return AMPBAM_EIO;
    const size_t used = std::min(w->pend.size(), nblk * BS), rest = w->pend.size() - used;
    if (rest) std::memmove(w->pend.data(), w->pend.data() + used, rest);       // < one block
    w->pend.n = rest;
    return AMPBAM_OK;
}

int ampbam_writer_open(const char *path, const char *header_text, int64_t header_len, const ampbam_file *like,
                       int level, int n_threads, ampbam_writer **out) {
    if (!path || !out || header_len < 0 || (header_len && !header_text) || !like || level < -1 || level > 9) return AMPBAM_EINVAL;
    *out = nullptr;
    ampbam_writer *w = new (std::nothrow) ampbam_writer();
    if (!w) return AMPBAM_ENOMEM;
    w->level = level; w->n_threads = pick_threads(n_threads);
    w->fp = std::fopen(path, "wb");
    if (!w->fp) { delete w; return AMPBAM_EIO; }
    Bytes &b = w->pend;
    uint8_t t[4];
    bool okm = append(b, "BAM\1", 4);
    put32(t, (uint32_t)header_len); okm = okm && append(b, t, 4);
    okm = okm && append(b, header_text, (size_t)header_len);
    put32(t, (uint32_t)like->ref_names.size()); okm = okm && append(b, t, 4);
    for (size_t r = 0; okm && r < like->ref_names.size(); ++r) {
        const std::string &nm = like->ref_names[r];
        put32(t, (uint32_t)nm.size() + 1); okm = okm && append(b, t, 4);
        okm = okm && append(b, nm.c_str(), nm.size() + 1);
        put32(t, (uint32_t)like->ref_lens[r]); okm = okm && append(b, t, 4);
    }
    if (!okm) { std::fclose(w->fp); delete w; return AMPBAM_ENOMEM; }
    // the header gets BGZF blocks of its own (htslib flushes behind it too): the files that the ranks of a multi-GPU run write can
    // then be joined block-wise, without the headers of the later ones (ampbam_writer_header_bytes)
    const int rcf = flush_blocks(w, true);
    if (rcf) { std::fclose(w->fp); delete w; return rcf; }
    w->header_bytes = (int64_t)std::ftell(w->fp);
    *out = w;
    return AMPBAM_OK;
}

int64_t ampbam_writer_header_bytes(const ampbam_writer *w) { return w ? w->header_bytes : -1; }

int ampbam_write_rows(ampbam_writer *w, const ampbam_file *src, int64_t n_rows, const int64_t *src_index,
                      const uint8_t *keep, const int32_t *new_pos, const uint32_t *new_ncig,
                      const uint64_t *new_cig_off, const uint32_t *new_cig) {
    if (!w || !src || n_rows < 0 || (n_rows && (!src_index || !keep || !new_pos || !new_ncig || !new_cig_off || !new_cig))) return AMPBAM_EINVAL;
    const uint8_t *d = src->data.data();
    const int64_t n_rec = ampbam_n_records(src);
    // sizes first, so that rows can be encoded in parallel straight into the pending buffer
    std::vector<uint64_t> off((size_t)n_rows + 1, 0);
    for (int64_t i = 0; i < n_rows; ++i) {
        uint64_t sz = 0;
        if (keep[i]) {
            const int64_t r = src_index[i];
            if (r < 0 || r >= n_rec || new_ncig[i] > 65535u) return AMPBAM_EINVAL;
            const uint64_t bs = src->rec_off[(size_t)r + 1] - src->rec_off[(size_t)r] - 4;
            sz = 4 + bs - 4ull * ((src->rec_info[(size_t)r] >> 32) & 0xFFFFu) + 4ull * new_ncig[i];   // no touch of the image here
        }
        off[(size_t)i + 1] = off[(size_t)i] + sz;
    }
    const size_t base = w->pend.size();
    if (!w->pend.resize(base + (size_t)off[(size_t)n_rows])) return AMPBAM_ENOMEM;
    uint8_t *ob = w->pend.data() + base;
    const int64_t grain = 4096;
    parallel_for(w->n_threads, (n_rows + grain - 1) / grain, [&](int64_t ch) {
        for (int64_t i = ch * grain; i < std::min(n_rows, (ch + 1) * grain); ++i) {
            if (!keep[i]) continue;
            const int64_t r = src_index[i];
            const uint8_t *c = d + src->rec_off[(size_t)r] + 4;
            const uint64_t bs = src->rec_off[(size_t)r + 1] - src->rec_off[(size_t)r] - 4;
            const uint32_t l_name = c[8], old_n = le16(c + 12), nn = new_ncig[i];
            uint8_t *o = ob + off[(size_t)i];
            put32(o, (uint32_t)(bs - 4ull * old_n + 4ull * nn));
            uint8_t *q = o + 4;
            std::memcpy(q, c, 32 + l_name);                                  // fixed fields + name
            const uint32_t *cg = new_cig + new_cig_off[i];
            int64_t rlen = 0;
            for (uint32_t k = 0; k < nn; ++k) {
                const uint32_t op = cg[k] & 15u;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cg[k] >> 4;
            }
            const int64_t pos = new_pos[i], end = pos + (rlen ? rlen : 1);
            put32(q + 4, (uint32_t)new_pos[i]);
            put16(q + 10, reg2bin(pos > 0 ? pos : 0, end > 1 ? end : 1));
            put16(q + 12, (uint16_t)nn);
            std::memcpy(q + 32 + l_name, cg, 4ull * nn);
            const uint64_t tail_from = 32ull + l_name + 4ull * old_n;          // bases, qualities, aux
            std::memcpy(q + 32 + l_name + 4ull * nn, c + tail_from, bs - tail_from);
        }
    });
    return flush_blocks(w, false);
}

int ampbam_write_batch(ampbam_writer *w, int64_t n, const int32_t *pos, const uint16_t *flag, const int32_t *tlen, const uint32_t *lseq,
                       const uint64_t *cig_off, const uint32_t *cig, const uint64_t *seq_off, const uint8_t *seq, const uint8_t *qual,
                       uint64_t name_base) {
    if (!w || n < 0 || (n && (!pos || !flag || !tlen || !lseq || !cig_off || !cig || !seq_off || !seq || !qual))) return AMPBAM_EINVAL;
    // sizes first (names are "r<number>"), so that rows can be encoded in parallel straight into the pending buffer
    auto digits = [](uint64_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; };
    std::vector<uint64_t> off((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t nc = cig_off[i + 1] - cig_off[i], L = lseq[i];
        if (nc > 65535u || (seq_off[i] & 1)) return AMPBAM_EINVAL;
        off[(size_t)i + 1] = off[(size_t)i] + 4 + 32 + (uint64_t)(2 + digits(name_base + (uint64_t)i)) + 4 * nc + (L + 1) / 2 + L;
    }
    const size_t base = w->pend.size();
    if (!w->pend.resize(base + (size_t)off[(size_t)n])) return AMPBAM_ENOMEM;
    uint8_t *ob = w->pend.data() + base;
    const int64_t grain = 4096;
    parallel_for(w->n_threads, (n + grain - 1) / grain, [&](int64_t ch) {
        for (int64_t i = ch * grain; i < std::min(n, (ch + 1) * grain); ++i) {
            uint8_t *o = ob + off[(size_t)i];
            const uint32_t bs = (uint32_t)(off[(size_t)i + 1] - off[(size_t)i] - 4), nc = (uint32_t)(cig_off[i + 1] - cig_off[i]), L = lseq[i];
            char name[24];
            const int ln = std::snprintf(name, sizeof(name), "r%llu", (unsigned long long)(name_base + (uint64_t)i)) + 1;
            const uint32_t *cg = cig + cig_off[i];
            int64_t rlen = 0;
            for (uint32_t k = 0; k < nc; ++k) { const uint32_t op = cg[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cg[k] >> 4; }
            const int64_t p = pos[i], end = p + (rlen ? rlen : 1);
            put32(o, bs);
            uint8_t *q = o + 4;
            put32(q, 0); put32(q + 4, (uint32_t)pos[i]);
            q[8] = (uint8_t)ln; q[9] = 60;
            put16(q + 10, reg2bin(p > 0 ? p : 0, end > 1 ? end : 1));
            put16(q + 12, (uint16_t)nc); put16(q + 14, flag[i]);
            put32(q + 16, L); put32(q + 20, 0); put32(q + 24, (uint32_t)pos[i]); put32(q + 28, (uint32_t)tlen[i]);
            std::memcpy(q + 32, name, (size_t)ln);
            std::memcpy(q + 32 + ln, cg, 4ull * nc);
            std::memcpy(q + 32 + ln + 4ull * nc, seq + seq_off[i] / 2, (L + 1) / 2);
            if (L & 1) q[32 + ln + 4ull * nc + L / 2] &= 0xF0;                       // (the pad nibble of an odd read is zero)
            std::memcpy(q + 32 + ln + 4ull * nc + (L + 1) / 2, qual + seq_off[i], L);
            if (L && qual[seq_off[i]] == 0xFF) std::memset(q + 32 + ln + 4ull * nc + (L + 1) / 2, 0xFF, L);      // QUAL '*'
        }
    });
    return flush_blocks(w, false);
}

int ampbam_writer_close(ampbam_writer *w) {
    if (!w) return AMPBAM_EINVAL;
    int rc = flush_blocks(w, true);
    if (rc == AMPBAM_OK && std::fwrite(BGZF_EOF, 1, sizeof(BGZF_EOF), w->fp) != sizeof(BGZF_EOF)) rc = AMPBAM_EIO;
    if (std::fclose(w->fp) != 0 && rc == AMPBAM_OK) rc = AMPBAM_EIO;
    delete w;
    return rc;
}

}  // extern "C"
