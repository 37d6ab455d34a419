// amp_fast.hpp -- the fast tile kernel (variant 4, the default): trim + pileup of SIMPLE reads in ONE pass
// over their bytes, written for CDNA4 / gfx950.
//
// A read is simple when its CIGAR is one match op covering the whole query ("150M", "150=", ...): nine
// reads in ten of an amplicon run.  For such a read every stage of trim_read (A:426-687) has a closed form
// (SimpleCig in amp_read.hpp: the result is always [S a][M m][S c]) and update_base_counts (A:690-753)
// reduces to "count base q at reference position pos' + (q - a) when qual[q] >= min_quality, a <= q < a+m".
// Everything else -- any other CIGAR, QUAL '*', reads too long for a tile -- goes on a list that the
// general tile kernel (amp_tile.hpp, k_tile<LIST>) processes afterwards with the exact generic code.
//
// Work decomposition:
//   * a block of F_WAVES waves owns a contiguous range of the coordinate-sorted batch and ONE LDS window of
//     per-position counters (win[6][F_W] uint32) anchored at its first read; positions outside it go
//     straight to the global table (results never depend on the input order).  Waves never synchronise
//     with each other inside the loop.
//   * a wave walks its share of the range in TILES of up to 32 reads whose quality / base bytes form one
//     contiguous run of at most F_CAP 8-base CHUNKS (reads start on 8-base boundaries):
//       A  lane = read   header fields, first CIGAR word, primer tables, closed-form primer clips
//                        (A:450-558); the tile's chunk loads are issued BEFORE the table look-ups are
//                        waited for: they depend only on the header
//       B  lane = chunk  slot s of lane l holds chunk 64 s + l of the run (coalesced 8-byte quality and
//                        4-byte base loads).  Owner read by popcount over a bit mask of read starts; the
//                        sliding-window scan (A:561-649) as in amp_tile.hpp; the per-base
//                        `quality >= min_quality` bits (8 per chunk) and the packed bases STAY IN REGISTERS
//       C  lane = read   quality clip (A:589-686), outputs, the read's counted query range
//       D  lane = chunk  counting from the registers of B: no byte of the batch is loaded twice
#pragma once

#include "amp_tile.hpp"

namespace amp {

constexpr int F_WAVES = 4;            // waves per block
constexpr int F_READS = 32;           // reads per tile
constexpr int F_NCH = 10;             // chunk slots per lane
constexpr int F_CAP = 64 * F_NCH;     // chunks per tile (5120 bases)
constexpr int F_W = 512;              // reference positions covered by the block's LDS window

struct FastWaveLds {
    uint2 info[F_READS];                 // x = lo | hi << 16 (aligned-quality window), y = coff | active << 30 | reverse << 31
    uint2 seg[F_READS];                  // x = qa | qb << 16 (counted query range), y = reference position of base qa - win_base
    uint32_t ff[F_READS];                // result of the window scan (S_FF encoding of amp_tile.hpp)
    uint32_t err[F_READS];               // bit 0: an uncountable base, bit 1: QUAL '*'
    unsigned long long smask[F_NCH];     // bit b of word s: a read starts at chunk 64 s + b
};
struct FastLds {
    uint32_t win[AMP_NSYM * F_W];
    uint32_t lut[16];
    uint32_t gcount;
    FastWaveLds wv[F_WAVES];
};

struct FastGrid { int64_t grid, rpb; };
static inline FastGrid fast_grid(int64_t n_reads) {
    // about 4096 blocks at most; a wave gets at least four tiles
    int64_t rpb = (n_reads + 4095) / 4096;
    rpb = ((rpb + F_WAVES * F_READS - 1) / (F_WAVES * F_READS)) * (F_WAVES * F_READS);
    if (rpb < 512) rpb = 512;
    return FastGrid{(n_reads + rpb - 1) / rpb, rpb};
}

// bits 7/15/23/31 of x -> bits 0..3
__device__ __forceinline__ uint32_t gather_msb4(uint32_t x) {
    uint32_t y = x >> 7;          // bits 0, 8, 16, 24
    y |= y >> 7;                  // bits 0,1 | 8,9 | 16,17 | 24
    y |= y >> 14;                 // bits 2,3 from 16,17
    return y & 15u;
}

// per-byte flag (bit 7) "quality >= mq" of four qualities, for mq <= 128
__device__ __forceinline__ uint32_t ok_bits4(uint32_t q, uint32_t mqb) {
    return ((((q & 0x7F7F7F7Fu) | 0x80808080u) - mqb) | q) & 0x80808080u;
}

// bit b of v8 -> bit 4b (one flag per base nibble)
__device__ __forceinline__ uint32_t spread8(uint32_t v8) {
    uint32_t x = (v8 | (v8 << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x;
}

// value of lane + 1 (lane 63: `last`): one DPP move on the VALU, no LDS crossbar
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t last, int lane) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
    return lane == 63 ? last : x;
}

template <int W>
__device__ __forceinline__ void fast_scan(const lds_u32 *smask32, const lds_u32 *info32, lds_u32 *ff, lds_u32 *err,
                                          const uint2 (&a0)[F_NCH], int lane, uint32_t T,
                                          uint32_t last_read, uint32_t thr, int32_t mq, uint32_t (&okm)[(F_NCH + 3) / 4],
                                          uint32_t (&own)[(F_NCH + 1) / 2]) {
    uint32_t base_cnt = 0;
#pragma unroll
    for (int s = 0; s < F_NCH; ++s) {
        if ((uint32_t)(s * 64) < T) {                                   // wave-uniform
            const uint32_t mlo = __builtin_amdgcn_readfirstlane(smask32[2 * s]);
            const uint32_t mhi = __builtin_amdgcn_readfirstlane(smask32[2 * s + 1]);
            const uint32_t c = (uint32_t)(s * 64 + lane);
            const bool live = c < T;
            // owner = (reads that start at or before chunk c) - 1
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            const uint32_t ownbit = (uint32_t)((((uint64_t)mhi << 32 | mlo) >> lane) & 1ull);
            uint32_t r = base_cnt + below + ownbit - 1u;
            r = r > last_read ? last_read : r;                          // lanes past the run (not live)
            base_cnt += (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi);
            const uint32_t ix = info32[2 * r], iy = info32[2 * r + 1];
            const int32_t rlo = (int32_t)(ix & 0xFFFFu), rhi = (int32_t)(ix >> 16);
            const uint32_t cj = c - (iy & 0xFFFFu);                     // chunk index inside the read
            const int32_t j0 = (int32_t)(cj * 8u);
            const bool act = live && ((iy >> 30) & 1u);
            // the next 8 qualities are the next lane's chunk (lane 63: lane 0 of the next slot); past the end of a
            // read they are another read's bytes, but no window that is looked at reaches them
            const uint2 q0 = a0[s];
            const uint2 nx = s + 1 < F_NCH ? a0[s + 1] : make_uint2(0u, 0u);
            uint2 q1;
            q1.x = wave_shl1(q0.x, __builtin_amdgcn_readfirstlane(nx.x), lane);
            q1.y = W > 4 ? wave_shl1(q0.y, __builtin_amdgcn_readfirstlane(nx.y), lane) : 0u;
            if (act) {
                uint32_t fail = window_fail_bits16<W>(q0, q1, thr);
                int32_t blo = rlo - j0, bhi = rhi - W - j0;             // window starts j0+b must lie in [rlo, rhi - W]
                blo = blo < 0 ? 0 : blo; bhi = bhi > 7 ? 7 : bhi;
                const uint32_t mk = (bhi < 0 || blo > 7) ? 0u : ((0xFFu >> (7 - bhi)) & (0xFFu << blo) & 0xFFu);
                fail &= mk;
                if (fail) {
                    const bool rv = (iy >> 31) != 0;
                    const uint32_t vf = (uint32_t)(j0 + (__builtin_ffs((int)fail) - 1) - rlo);            // first failing window start
                    const uint32_t vr = 0xFFFFu - (uint32_t)(j0 + (31 - __builtin_clz(fail)) + W - rlo);  // last failing window end
                    __hip_atomic_fetch_min(ff + r, rv ? vr : vf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (live && cj == 0u && (q0.x & 0xFFu) == 0xFFu)            // QUAL '*' marker of the owner
                __hip_atomic_fetch_or(err + r, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t k8 = 0;
            if (live) {
                uint32_t ok0, ok1;
                if (mq <= 128) {
                    const uint32_t mqb = (uint32_t)mq * 0x01010101u;
                    ok0 = ok_bits4(q0.x, mqb); ok1 = ok_bits4(q0.y, mqb);
                } else {
                    ok0 = ok1 = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        ok0 |= (((q0.x >> (8 * b)) & 0xFFu) >= (uint32_t)mq ? 0x80u : 0u) << (8 * b);
                        ok1 |= (((q0.y >> (8 * b)) & 0xFFu) >= (uint32_t)mq ? 0x80u : 0u) << (8 * b);
                    }
                }
                k8 = gather_msb4(ok0) | (gather_msb4(ok1) << 4);
            }
            if ((s & 3) == 0) okm[s >> 2] = k8; else okm[s >> 2] |= k8 << (8 * (s & 3));
            const uint32_t m16 = r | (cj << 5);                         // owner (5 bits) | chunk in read (11 bits)
            if ((s & 1) == 0) own[s >> 1] = m16; else own[s >> 1] |= m16 << 16;
        } else {
            if ((s & 3) == 0) okm[s >> 2] = 0;
            if ((s & 1) == 0) own[s >> 1] = 0;
        }
    }
}

__global__ void __launch_bounds__(F_WAVES * 64, 4)
k_fast(KParams P, amp_dev_reads rd, DevOut out, uint32_t *counts, unsigned long long *ctr, uint32_t *glist, uint32_t *gcnt,
       int reads_per_block) {
    __shared__ FastLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t n = rd.n_reads;
    const int64_t rb = (int64_t)blockIdx.x * reads_per_block;
    const int64_t re = rb + reads_per_block < n ? rb + reads_per_block : n;
    lds_u32 *const win = (lds_u32 *)L.win;
    lds_u32 *const lut = (lds_u32 *)L.lut;
    for (int i = tid; i < AMP_NSYM * F_W; i += F_WAVES * 64) win[i] = 0;
    if (tid == 0) L.gcount = 0;
    if (tid < 16) { uint32_t c = col_of_code((uint32_t)tid); lut[tid] = c <= 4u ? c * (uint32_t)(F_W * 4) : 0u; }
    FastWaveLds &Wv = L.wv[wave];
    lds_u32 *const info32 = (lds_u32 *)Wv.info;
    lds_u32 *const seg32 = (lds_u32 *)Wv.seg;
    lds_u32 *const ff = (lds_u32 *)Wv.ff;
    lds_u32 *const err = (lds_u32 *)Wv.err;
    lds_u32 *const smask32 = (lds_u32 *)Wv.smask;
    if (lane < 2 * F_NCH) smask32[lane] = 0;
    // the block's window: anchored at its first read (sorted input: nothing of this block starts left of it)
    int32_t win_base = rb < n ? rd.pos[rb] : 0;
    win_base = (win_base < 0 ? 0 : win_base) & ~31;
    uint32_t wlim;
    {
        const int64_t lim = (int64_t)P.ref_len - win_base;
        wlim = lim <= 0 ? 0u : (lim > F_W ? (uint32_t)F_W : (uint32_t)lim);
    }
    __syncthreads();

    const int32_t mq = P.min_quality, Wd = P.window;
    const uint32_t mqc = (uint32_t)(mq > 256 ? 256 : mq);
    const uint32_t thr = mqc * (uint32_t)Wd;
    const uint32_t G = (uint32_t)P.ref_len;
    const int64_t per_wave = reads_per_block / F_WAVES;
    const int64_t wbeg = rb + (int64_t)wave * per_wave;
    int64_t wend = wbeg + per_wave;
    wend = wend < re ? wend : re;
    unsigned long long n_err = 0;
    const uint32_t rot = ((uint32_t)lane >> 2) & 7u;

    for (int64_t i0 = wbeg; i0 < wend;) {
        // =================================== A: lane = read ===================================
        const int64_t i = i0 + lane;
        const bool valid = lane < F_READS && i < wend;
        int32_t pos = 0, tlen = 0;
        uint32_t lseq = 0, flag = 0, c0 = 0, c1 = 0, o8 = 0;
        if (valid) {
            pos = rd.pos[i]; flag = rd.flag[i]; tlen = rd.tlen[i]; lseq = rd.lseq[i];
            c0 = rd.cig_off32[i]; c1 = rd.cig_off32[i + 1]; o8 = rd.seq_off8[i];
        }
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(o8);
        const uint32_t nch = (lseq + 7u) >> 3;
        const uint32_t coff = o8 - m0;
        const bool fits = valid && o8 >= m0 && lseq > 0u && lseq < 65536u && coff <= (uint32_t)F_CAP && coff + nch <= (uint32_t)F_CAP;
        const unsigned long long fitmask = __ballot(fits);
        int ntake = fitmask == ~0ull ? 64 : __builtin_ctzll(~fitmask);   // leading reads that fit the tile's run
        const bool solo = ntake == 0;                                     // the first read alone is too long: general pass
        if (solo) ntake = 1;
        const bool taken = lane < ntake;
        const uint32_t last_read = (uint32_t)(ntake - 1);
        const uint32_t T = solo ? 0u : (uint32_t)__shfl((int)(coff + nch), ntake - 1);   // chunks of the run

        // second level of loads, all issued before any is waited for.  First what the read lanes need next
        // (first CIGAR word, the two primer-table entries of A:450-451: for a simple read they depend on the
        // header only), then the tile's bytes, which are consumed in B: vmcnt retires in order, so the wait
        // for the former leaves the latter in flight.
        uint32_t w0 = 0;
        if (taken && !solo && c1 > c0) w0 = rd.cig[c0];
        const bool in_ref = (uint32_t)pos < G && (uint32_t)(pos + (int32_t)lseq - 1) < G;        // A:450-451
        int32_t tabL = -1, tabR = -1;
        if (taken && !solo && P.do_trim && in_ref) { tabL = P.max_end[pos]; tabR = P.min_start[pos + (int32_t)lseq - 1]; }
        // Straight-line loads (lanes past the run re-read its last chunk): a branch per slot made the compiler
        // wait for every load before issuing the next.
        uint2 a0[F_NCH];
        uint32_t sw[F_NCH];
        {
            const uint32_t tm1 = T ? T - 1u : 0u;
#pragma unroll
            for (int s = 0; s < F_NCH; ++s) {
                const uint32_t cl = (uint32_t)(s * 64 + lane) < tm1 ? (uint32_t)(s * 64 + lane) : tm1;
                a0[s] = *(const uint2 *)(rd.qual + ((int64_t)m0 + cl) * 8);
                sw[s] = *(const uint32_t *)(rd.seq + ((int64_t)m0 + cl) * 4);
            }
        }
        const bool simple = taken && !solo && c1 - c0 == 1u && is_simple_cigar(1, w0, (int32_t)lseq);
        const bool rev = (flag & 0x10u) != 0;
        TrimState ts{pos, 1, 0u, 0};
        SimpleCig sc{w0 & 15u, 0, (int32_t)lseq, 0};
        if (simple && P.do_trim) {
            if (!in_ref) ts.err = AMP_RS_INDEX_REF;
            else trim_primers_simple_tab(P, ts, flag, tlen, (int32_t)lseq, sc, tabL, tabR);
        }
        const bool scan = simple && P.do_trim && !ts.err;
        const int32_t lo = sc.m > 0 ? sc.a : (int32_t)lseq, qlen = sc.m, hi = lo + qlen;
        // the 3' end's shrinking windows (A:575-576, A:637-638) need at most W-1 bytes: fetched now
        // (loaded by every lane, from the start of its read when there is nothing to scan: a load under a
        // branch is waited for at the end of the branch, and this one is not needed before C)
        const int32_t first = !scan ? 0 : (rev || qlen < Wd) ? lo : lo + qlen - Wd + 1;
        const int32_t tab = first & ~7;
        const uint8_t *twp = rd.qual + (int64_t)o8 * 8 + tab;
        const uint2 tw0 = *(const uint2 *)twp, tw1 = *(const uint2 *)(twp + 8);
        if (lane < F_READS) {
            info32[2 * lane] = scan ? ((uint32_t)lo | ((uint32_t)hi << 16)) : 0u;
            info32[2 * lane + 1] = (taken ? coff : 0u) | (scan ? 1u << 30 : 0u) | (rev ? 1u << 31 : 0u);
            ff[lane] = 0xFFFFu;
            err[lane] = 0u;
        }
        if (taken && !solo) {
            // 32-bit halves of the start mask (ds_or_b32: two reads may start in the same word)
            __hip_atomic_fetch_or(smask32 + ((coff >> 6) * 2u + ((coff >> 5) & 1u)), 1u << (coff & 31u), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_sync();

        // =================================== B: lane = chunk ===================================
        uint32_t okm[(F_NCH + 3) / 4], own[(F_NCH + 1) / 2];
        switch (Wd) {
            case 1: fast_scan<1>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            case 2: fast_scan<2>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            case 3: fast_scan<3>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            case 4: fast_scan<4>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            case 5: fast_scan<5>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            case 6: fast_scan<6>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            case 7: fast_scan<7>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
            default: fast_scan<8>(smask32, info32, ff, err, a0, lane, T, last_read, thr, mq, okm, own); break;
        }
        wave_sync();

        // =================================== C: lane = read ===================================
        bool general = taken && !simple;          // solo included
        bool counted = false;
        if (simple) {
            const bool no_qual = (err[lane] & 2u) != 0;
            if (no_qual) {
                general = true;                   // QUAL '*': the generic code reports it (A:561-562, A:718)
            } else {
                if (scan) {
                    int32_t iq;
                    const uint32_t v = ff[lane];
                    if (v != 0xFFFFu) {
                        iq = rev ? (int32_t)(0xFFFFu - v) : (int32_t)v;
                    } else {
                        // no full window failed: the shrinking windows at the 3' end decide
                        iq = rev ? 0 : qlen;
                        int32_t acc = 0;
                        const uint64_t t_lo = (uint64_t)tw0.x | ((uint64_t)tw0.y << 32), t_hi = (uint64_t)tw1.x | ((uint64_t)tw1.y << 32);
                        const int32_t kmax = qlen < Wd - 1 ? qlen : Wd - 1;
                        for (int32_t k = 1; k <= kmax; ++k) {
                            const uint32_t o = (uint32_t)((rev ? lo + k - 1 : hi - k) - tab);      // 0..15
                            acc += (int32_t)(((o < 8u ? t_lo : t_hi) >> ((o & 7u) * 8u)) & 0xFFu);
                            if ((int64_t)acc < (int64_t)mq * k) iq = rev ? k : qlen - k;
                        }
                    }
                    trim_quality_apply_simple(ts, rev, iq, qlen, sc);
                }
                const size_t slot = (size_t)c0 + 3 * (size_t)i;
                uint32_t ncig = 0;
                int32_t reflen = 0;
                if (!ts.err) {
                    uint32_t *home = out.new_cig + slot;
                    if (sc.a > 0) home[ncig++] = ((uint32_t)sc.a << 4) | OP_S;
                    if (sc.m > 0) home[ncig++] = ((uint32_t)sc.m << 4) | sc.op;
                    if (sc.c > 0) home[ncig++] = ((uint32_t)sc.c << 4) | OP_S;
                    reflen = sc.m > 0 ? sc.m : 1;
                }
                if (out.new_pos) out.new_pos[i] = ts.pos;
                if (out.new_ncig) out.new_ncig[i] = ncig;
                if (out.ref_len) out.ref_len[i] = reflen;
                if (out.trim_flags) out.trim_flags[i] = ts.err ? (uint8_t)0 : (uint8_t)ts.flags;
                if (out.status) out.status[i] = (uint8_t)ts.err;
                if (ts.err) ++n_err;
                counted = !ts.err && P.do_count;
            }
        }
        if (lane < F_READS) {
            seg32[2 * lane] = counted ? ((uint32_t)sc.a | ((uint32_t)(sc.a + sc.m) << 16)) : 0u;
            seg32[2 * lane + 1] = (uint32_t)(ts.pos - win_base);
        }
        wave_sync();

        // =================================== D: lane = chunk ===================================
        if (P.do_count) {
#pragma unroll
            for (int s = 0; s < F_NCH; ++s) {
                if ((uint32_t)(s * 64) < T) {                               // wave-uniform
                    const uint32_t k8 = (okm[s >> 2] >> (8 * (s & 3))) & 0xFFu;
                    const uint32_t m16 = (own[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
                    const uint32_t r = m16 & 31u;
                    const int32_t j0 = (int32_t)((m16 >> 5) * 8u);
                    const uint32_t sx = seg32[2 * r], sy = seg32[2 * r + 1];
                    const int32_t m0q = (int32_t)(sx & 0xFFFFu), m1q = (int32_t)(sx >> 16);
                    int32_t klo = m0q - j0, khi = m1q - j0;
                    klo = klo < 0 ? 0 : klo; khi = khi > 8 ? 8 : khi;
                    const uint32_t rng = khi > klo ? ((1u << khi) - 1u) & ~((1u << klo) - 1u) : 0u;
                    const uint32_t v8 = k8 & rng;                           // bases of this chunk that are counted
                    if (v8) {
                        const int32_t d0 = (int32_t)sy + (j0 - m0q);        // window offset of base 0 of the chunk
                        uint32_t x = sw[s];
                        x = ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);   // base k at bits [4k, 4k+4)
                        uint32_t pc = x - ((x >> 1) & 0x55555555u);
                        pc = (pc & 0x33333333u) + ((pc >> 2) & 0x33333333u);       // per-nibble popcount
                        const uint32_t good = (pc ^ (pc >> 2)) & ~(pc >> 1) & 0x11111111u;   // popcount 1 (A C G T) or 4 (N)
                        const bool safe = (~good & spread8(v8)) == 0u && wlim >= 8u && (uint32_t)d0 <= wlim - 8u;
                        if (safe) {
                            // the 8 bases rotated by `rot`: lanes serviced together spread over the banks
                            const uint32_t vr = ((v8 | (v8 << 8)) >> rot) & 0xFFu;
                            const uint32_t sr = __builtin_amdgcn_alignbit(x, x, rot * 4u);
                            lds_u8 *const wbase = (lds_u8 *)win + (uint32_t)d0 * 4u;
                            uint32_t plane[8];
#pragma unroll
                            for (int b = 0; b < 8; ++b) {
                                const uint32_t code4 = b == 0 ? (sr << 2) & 0x3Cu : (sr >> (4 * b - 2)) & 0x3Cu;
                                plane[b] = *(lds_u32 *)((lds_u8 *)lut + code4);
                            }
#pragma unroll
                            for (int b = 0; b < 8; ++b) {
                                const uint32_t bb4 = ((rot + (uint32_t)b) & 7u) * 4u;
                                lds_add((lds_u32 *)(wbase + plane[b] + bb4), (vr >> b) & 1u);
                            }
                        } else {
                            bool bad = false;
#pragma unroll 1
                            for (int b = 0; b < 8; ++b) {
                                if (!((v8 >> b) & 1u)) continue;
                                const uint32_t col = col_of_code((x >> (4 * b)) & 15u);
                                const int32_t rp = win_base + d0 + b;
                                const uint32_t d = (uint32_t)(d0 + b);
                                if (col > 4u || (uint32_t)rp >= G) bad = true;
                                else if (d < (uint32_t)F_W) lds_add(win + col * F_W + d, 1u);
                                else atomicAdd(&counts[(size_t)rp * AMP_NSYM + col], 1u);
                            }
                            if (bad) err[r] = 1u;
                        }
                    }
                }
            }
        }
        wave_sync();

        // ---- hand-over to the general pass (lane = read): one reservation per wave ------------------
        {
            uint32_t entry = 0;
            bool has = false;
            if (general) { entry = (uint32_t)i; has = true; }
            else if (counted && (err[lane] & 1u)) { entry = (uint32_t)i | GL_STATUS_ONLY; has = true; }
            const unsigned long long m = __ballot(has);
            if (m) {
                uint32_t gb = 0;
                if (lane == 0) gb = __hip_atomic_fetch_add((lds_u32 *)&L.gcount, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                gb = __shfl(gb, 0);
                if (has) glist[(size_t)rb + gb + __popcll(m & ((1ull << lane) - 1ull))] = entry;
            }
            if (lane < 2 * F_NCH) smask32[lane] = 0;
        }
        wave_sync();
        i0 += ntake;
    }

    __syncthreads();
    for (int i = tid; i < AMP_NSYM * F_W; i += F_WAVES * 64) {
        const uint32_t v = win[i];
        if (v) {
            const int sym = i / F_W, d = i - sym * F_W;
            atomicAdd(&counts[(size_t)(win_base + d) * AMP_NSYM + sym], v);
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
    if (tid == 0) gcnt[blockIdx.x] = L.gcount;
}

// Dense list of the reads the fast kernel handed over + the geometry of the general pass.  Block b places
// its segment behind the totals of the blocks before it (the per-block counts are a few KB in L2).
__global__ void __launch_bounds__(256)
k_gcompact(const uint32_t *__restrict__ glist, const uint32_t *__restrict__ gcnt, int reads_per_block, uint32_t *__restrict__ dense,
           GenGeo *geo, uint32_t gen_grid) {
    __shared__ uint32_t s_part[4];
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) acc += gcnt[b];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    const uint32_t off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const uint32_t cnt = gcnt[blockIdx.x];
    const uint32_t *src = glist + (size_t)blockIdx.x * (size_t)reads_per_block;
    for (uint32_t k = tid; k < cnt; k += 256) dense[off + k] = src[k];
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const uint32_t n_list = off + cnt;
        const uint32_t tiles = (n_list + TILE - 1) / TILE;
        uint32_t tpb = (tiles + gen_grid - 1) / gen_grid;
        tpb = ((tpb + T_WAVES - 1) / T_WAVES) * T_WAVES;
        if (tpb < (uint32_t)T_WAVES) tpb = T_WAVES;
        geo->n_list = n_list; geo->tpb = tpb; geo->n_seg = (tiles + tpb - 1) / tpb; geo->pad = 0;
    }
}

}  // namespace amp
