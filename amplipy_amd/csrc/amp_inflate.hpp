// amp_inflate.hpp -- a DEFLATE (RFC 1951) decoder for BGZF blocks, written for libampbam.
//
// A BGZF member holds at most 64 KiB of raw DEFLATE whose inflated size is known in advance (ISIZE) and whose CRC-32 is
// checked afterwards, so the decoder can be specialised: one input buffer, one output buffer of exactly the expected size, no
// streaming state, no window (the output IS the window).  zlib 1.2.11's inflate does 0.4-0.6 GB/s per thread on BAM data and,
// after the block CRC went to carry-less multiplication, was all of the codec's inflate stage; this one keeps 56+ bits in a
// 64-bit buffer (one unaligned 8-byte refill per symbol pair instead of a byte at a time), decodes literal / length codes with
// a 10-bit first-level table and distance codes with an 8-bit one (second-level tables behind pointer entries for the rare
// longer codes) and copies matches eight bytes at a time where source and destination are far enough apart.
//
// Safety: every write is checked against the end of the output, every match distance against the bytes produced so far, the
// input pointer never passes the end of the input (missing bits read as zeros and are counted: consuming more bits than
// the input holds is an error).  A block this decoder refuses, or whose CRC then differs, is simply inflated again by zlib
// (ampbam.cpp), so a defect here can cost time but not correctness.  tests/test_bam_native.py runs it against zlib on
// streams of every compression level and strategy, on truncated and on random input.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace ampinf {

constexpr int LIT_TB = 10;            // first-level bits of the literal / length table
constexpr int DST_TB = 8;             // ... of the distance table
constexpr int PRE_TB = 7;             // the code-length code is at most 7 bits long: one level
constexpr int MAX_LEN = 15;
constexpr int N_LITLEN = 288, N_DIST = 32, N_PRE = 19;

// table entry: bits 0-4 code length (second level: the code's full length), bits 5-7 kind, bits 8-12 extra bits,
// bits 16-31 value (literal, base length, base distance, or index of a second-level table with its width in `extra`)
enum : uint32_t { K_LITERAL = 0u << 5, K_LENGTH = 1u << 5, K_EOB = 2u << 5, K_SUB = 3u << 5, K_INVALID = 4u << 5, K_MASK = 7u << 5 };
static inline uint32_t mk(uint32_t len, uint32_t kind, uint32_t extra, uint32_t value) { return len | kind | (extra << 8) | (value << 16); }

struct Tables {
    uint32_t lit[(1 << LIT_TB) + 1024];        // worst case of second-level entries for 288 symbols of up to 15 bits: < 1024
    uint32_t dst[(1 << DST_TB) + 512];
    uint32_t pre[1 << PRE_TB];
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t bit_reverse(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// what symbol `sym` of alphabet `which` (0 literal / length, 1 distance, 2 code lengths) decodes to
static inline uint32_t symbol_entry(int which, int sym, uint32_t len) {
    if (which == 2) return mk(len, K_LITERAL, 0, (uint32_t)sym);
    if (which == 1) return sym < 30 ? mk(len, K_LENGTH, DST_EXTRA[sym], DST_BASE[sym]) : mk(len, K_INVALID, 0, 0);
    if (sym < 256) return mk(len, K_LITERAL, 0, (uint32_t)sym);
    if (sym == 256) return mk(len, K_EOB, 0, 0);
    return sym < 286 ? mk(len, K_LENGTH, LEN_EXTRA[sym - 257], LEN_BASE[sym - 257]) : mk(len, K_INVALID, 0, 0);
}

// Canonical Huffman decoding table from code lengths (RFC 1951 3.2.2).  tb = first-level bits, cap = entries available.
// Returns false for an over-subscribed code; an incomplete code is accepted (unused patterns decode to K_INVALID), as zlib
// accepts the single-code distance alphabets real encoders emit.
static inline bool build_table(int which, const uint8_t *lens, int n_sym, uint32_t *tab, int tb, int cap) {
    int count[MAX_LEN + 1] = {0};
    for (int s = 0; s < n_sym; ++s) ++count[lens[s]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= MAX_LEN; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
    uint32_t next_code[MAX_LEN + 2];
    uint32_t code = 0;
    for (int l = 1; l <= MAX_LEN; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
    const int first = 1 << tb;
    for (int i = 0; i < first; ++i) tab[i] = mk(1, K_INVALID, 0, 0);
    // widths of the second-level tables: the longest code behind each first-level prefix
    uint8_t sub_bits[1 << LIT_TB];
    memset(sub_bits, 0, (size_t)first);
    {
        uint32_t nc[MAX_LEN + 2];
        memcpy(nc, next_code, sizeof(nc));
        for (int s = 0; s < n_sym; ++s) {
            const int l = lens[s];
            if (l <= tb) { if (l) ++nc[l]; continue; }
            const uint32_t rev = bit_reverse(nc[l]++, l);
            const uint32_t pfx = rev & (uint32_t)(first - 1);
            if (l - tb > sub_bits[pfx]) sub_bits[pfx] = (uint8_t)(l - tb);
        }
    }
    int used = first;
    for (int pfx = 0; pfx < first; ++pfx) {
        if (!sub_bits[pfx]) continue;
        const int n = 1 << sub_bits[pfx];
        if (used + n > cap) return false;
        tab[pfx] = mk((uint32_t)tb, K_SUB, sub_bits[pfx], (uint32_t)used);
        for (int i = 0; i < n; ++i) tab[used + i] = mk(1, K_INVALID, 0, 0);
        used += n;
    }
    for (int s = 0; s < n_sym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = bit_reverse(next_code[l]++, l);
        const uint32_t e = symbol_entry(which, s, (uint32_t)l);
        if (l <= tb) {
            for (uint32_t i = rev; i < (uint32_t)first; i += 1u << l) tab[i] = e;
        } else {
            const uint32_t p = tab[rev & (uint32_t)(first - 1)];
            const uint32_t start = p >> 16, bits = (p >> 8) & 31u;
            for (uint32_t i = rev >> tb; i < (1u << bits); i += 1u << (l - tb)) tab[start + i] = e;
        }
    }
    return true;
}

struct Bits {
    const uint8_t *in, *end;
    uint64_t buf = 0;
    int cnt = 0;                 // valid bits in buf
    int64_t phantom = 0;         // zero bits supplied behind the end of the input
    inline void refill() {
        if (end - in >= 8) {
            uint64_t w;
            memcpy(&w, in, 8);                       // little-endian host (x86-64; ampbam.cpp already assumes it for the CIGAR words)
            buf |= w << cnt;
            in += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) {
                if (in < end) buf |= (uint64_t)*in++ << cnt; else phantom += 8;
                cnt += 8;
            }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    inline bool overrun() const { return phantom > (int64_t)cnt; }      // bits beyond the input have been consumed
};

static inline const Tables &fixed_tables() {
    static const Tables T = [] {
        Tables t;
        uint8_t l[N_LITLEN];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        (void)build_table(0, l, N_LITLEN, t.lit, LIT_TB, (int)(sizeof(t.lit) / 4));
        uint8_t d[N_DIST];
        for (int i = 0; i < N_DIST; ++i) d[i] = 5;
        (void)build_table(1, d, N_DIST, t.dst, DST_TB, (int)(sizeof(t.dst) / 4));
        return t;
    }();
    return T;
}

// Inflates exactly out_len bytes from the raw DEFLATE stream in[0, in_len).  scratch: one Tables per thread.
// Returns true when the stream ended with its final block and produced exactly out_len bytes.
static inline bool inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, Tables &scratch) {
    Bits b{in, in + in_len};
    uint8_t *o = out, *const o_end = out + out_len;
    static const uint8_t PRE_ORDER[N_PRE] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        b.refill();
        const uint32_t final = b.take(1), type = b.take(2);
        const Tables *T;
        if (type == 0) {
            b.drop(b.cnt & 7);                                   // to the byte boundary
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xFFFFu) != nlen || b.overrun()) return false;
            // the bytes still in the bit buffer first (whole bytes: cnt is a multiple of 8 here), then straight from the input
            uint32_t left = len;
            if ((size_t)(o_end - o) < left) return false;
            while (left && b.cnt >= 8) {
                if ((int64_t)b.cnt - b.phantom < 8) return false;          // (that byte would be one of the zeros behind the input)
                *o++ = (uint8_t)b.take(8); --left;
            }
            if (left) {
                if (b.cnt != 0 || (size_t)(b.end - b.in) < left) return false;
                memcpy(o, b.in, left);
                b.in += left; o += left;
                b.buf = 0;                                       // (bits of the bytes just skipped may sit above cnt: see refill)
            }
            if (final) break;
            continue;
        } else if (type == 1) {
            T = &fixed_tables();
        } else if (type == 2) {
            const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            uint8_t pl[N_PRE] = {0};
            for (uint32_t i = 0; i < hclen; ++i) { if (b.cnt < 3) b.refill(); pl[PRE_ORDER[i]] = (uint8_t)b.take(3); }
            if (!build_table(2, pl, N_PRE, scratch.pre, PRE_TB, 1 << PRE_TB)) return false;
            uint8_t lens[N_LITLEN + N_DIST];
            uint32_t i = 0;
            while (i < hlit + hdist) {
                b.refill();
                const uint32_t e = scratch.pre[b.peek(PRE_TB)];
                if ((e & K_MASK) != K_LITERAL) return false;
                b.drop((int)(e & 31u));
                const uint32_t sym = e >> 16;
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + b.take(2); }
                else if (sym == 17) rep = 3 + b.take(3);
                else rep = 11 + b.take(7);
                if (i + rep > hlit + hdist) return false;
                memset(lens + i, (int)val, rep);
                i += rep;
            }
            if (b.overrun() || lens[256] == 0) return false;
            if (!build_table(0, lens, (int)hlit, scratch.lit, LIT_TB, (int)(sizeof(scratch.lit) / 4))) return false;
            if (!build_table(1, lens + hlit, (int)hdist, scratch.dst, DST_TB, (int)(sizeof(scratch.dst) / 4))) return false;
            T = &scratch;
        } else {
            return false;
        }
        // ---- the symbols of a compressed block ----
        for (;;) {
            b.refill();                                          // >= 56 bits: a length code, its extra bits, a distance code and its extra bits are <= 48
            uint32_t e = T->lit[b.peek(LIT_TB)];
            if ((e & K_MASK) == K_SUB) e = T->lit[(e >> 16) + ((uint32_t)(b.buf >> LIT_TB) & ((1u << ((e >> 8) & 31u)) - 1u))];
            b.drop((int)(e & 31u));
            const uint32_t kind = e & K_MASK;
            if (kind == K_LITERAL) {
                if (o >= o_end) return false;
                *o++ = (uint8_t)(e >> 16);
                // a second literal from the same refill (literals are at most 15 bits: 56 - 15 leaves room)
                uint32_t e2 = T->lit[b.peek(LIT_TB)];
                if ((e2 & K_MASK) == K_LITERAL && o < o_end) { b.drop((int)(e2 & 31u)); *o++ = (uint8_t)(e2 >> 16); }
                continue;
            }
            if (kind == K_EOB) break;
            if (kind != K_LENGTH) return false;
            const uint32_t len = (e >> 16) + b.take((int)((e >> 8) & 31u));
            uint32_t d = T->dst[b.peek(DST_TB)];
            if ((d & K_MASK) == K_SUB) d = T->dst[(d >> 16) + ((uint32_t)(b.buf >> DST_TB) & ((1u << ((d >> 8) & 31u)) - 1u))];
            if ((d & K_MASK) != K_LENGTH) return false;
            b.drop((int)(d & 31u));
            const uint32_t dist = (d >> 16) + b.take((int)((d >> 8) & 31u));
            if (dist > (size_t)(o - out) || len > (size_t)(o_end - o)) return false;
            const uint8_t *s = o - dist;
            if (dist >= 8 && (size_t)(o_end - o) >= (size_t)len + 8) {
                uint8_t *const stop = o + len;                    // eight bytes at a time, up to seven bytes of over-copy inside the buffer
                do { uint64_t w; memcpy(&w, s, 8); memcpy(o, &w, 8); s += 8; o += 8; } while (o < stop);
                o = stop;
            } else if (dist == 1) {
                memset(o, *s, len); o += len;
            } else {
                for (uint32_t k = 0; k < len; ++k) o[k] = s[k];
                o += len;
            }
        }
        if (b.overrun()) return false;
        if (final) break;
    }
    return o == o_end && !b.overrun();
}

}  // namespace ampinf
