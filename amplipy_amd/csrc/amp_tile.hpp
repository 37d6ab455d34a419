// amp_tile.hpp -- the fused trim + pileup kernel (variant 2), written for CDNA4 / gfx950.
//
// Work decomposition (nothing like the reference's per-read Python loop, A:896-915):
//   * a wave owns a TILE of 64 consecutive reads; a block of T_WAVES waves (two blocks per CU, the
//     grid about eight times oversubscribed) walks a contiguous range of tiles of the
//     coordinate-sorted batch, so it touches a bounded reference window
//   * per-position counters are PRIVATISED in LDS: win[6][W] uint32 for reference positions
//     [win_base, win_base+W); lanes add with LDS atomics (ds_add_u32) and the block flushes
//     the non-zero counters with global atomics when the window has to move / at the end.
//     Anything outside the window goes straight to the global table, so results never depend
//     on the input order -- only the speed does.
//   * the tile alternates between two lane mappings:
//       lane = read   P1  primer clips on the CIGAR held in LDS (one column per lane)
//                     P3  quality clip (partial windows + CIGAR rewrite), outputs, match-op
//                         SEGMENTS
//       lane = chunk  P2  sliding-window quality scan: a chunk is 8 consecutive window START
//                         positions; eight W-byte sums (v_sad_u8) from a 16-byte neighbourhood,
//                         first / last failing full window per read by one LDS atomicMin
//                     P4  base counting: a chunk is 8 bases of one match-op segment (8 B of
//                         qual + 4 B of packed seq, coalesced); straight-line code, one
//                         unconditional LDS add of 0/1 per base.  The base order inside a chunk
//                         is rotated per lane so that the 32 lanes serviced together hit 32
//                         different banks.
//     Chunk lanes find their owner through a byte map in LDS that the read lanes fill
//     (chunk -> read for P2, chunk -> segment for P4); the map aliases the spare CIGAR buffer.
//   * everything else is DEFERRED through the block's own segment of a list to the second pass in
//     amplihip.hip: "light" entries (deletions and insertion events of reads whose match bases were
//     counted here) to k_deferred_light, "heavy" entries to k_deferred_heavy, which runs the exact code
//     of amp_read.hpp -- reads with more CIGAR ops than the LDS columns hold, reads of 64 k bases or
//     more, reads whose trimmed CIGAR is not regular (clips only at the ends, body of M/=/X/I/D/N),
//     reads whose segments do not fit the tile's segment table, and (status only) reads on which a
//     chunk lane met an error.
#pragma once

#include "amp_read.hpp"

namespace amp {

constexpr int TILE = 64;          // reads per wave tile
constexpr int T_WAVES = 8;        // waves per block (two blocks per CU)
constexpr int T_W = 512;          // reference positions covered by the LDS window
constexpr int T_MAXOPS = 18;      // CIGAR ops per read held in LDS as 16-bit words (input ops <= T_MAXOPS-3, lengths sum < 4096)
constexpr int T_MAPCAP = T_MAXOPS * TILE * 2;   // chunk-map bytes = the spare CIGAR buffer
constexpr int T_SEGCAP = 192;     // match-op segments per tile
constexpr int T_UNROLL = 3;       // chunks per lane whose loads are issued before any of them is processed (P2)
constexpr int T_UNROLL4 = 3;      // ... in P4 (measured: 3 / 3 runs without register spills, 4 / 4 spills)
constexpr int32_t NO_WINDOW = INT32_MIN;

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef uint32_t amp_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) amp_u32x2 lds_u32x2;
typedef uint32_t amp_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) amp_u32x4 lds_u32x4;

// entries of the deferred list: read index | kind
constexpr uint32_t DEFER_STATUS_ONLY = 0x80000000u;   // counted by the tile kernel; only the exact status is missing
constexpr uint32_t DEFER_INDELS = 0x40000000u;        // match bases counted by the tile kernel; deletions / insertion events missing
constexpr uint32_t DEFER_INDEX_MASK = 0x3FFFFFFFu;

// The general pass of variant 4 (amp_fast.hpp) runs this kernel over a LIST of reads: entries are read index | kind
constexpr uint32_t GL_STATUS_ONLY = 0x80000000u;   // counted by the fast kernel; a base could not be counted: exact status wanted
constexpr uint32_t GL_LONG = 0x40000000u;          // tens of CIGAR ops: taken by k_long (amp_wave.hpp), not a row of this pass
constexpr uint32_t GL_INDEX_MASK = 0x3FFFFFFFu;
// Where k_tile<LIST> finds its list when k_gcompact has not packed it (the common case: one launch less per batch).  The fast
// kernel leaves one list segment per block (entries [b * rpb, b * rpb + gcnt[b]) of glist); a block of the tile kernel sums
// the counts itself (a KB from L2), derives the geometry k_gcompact would have written, and finds entry li of the virtual
// dense list by a binary search over the prefix sums.
constexpr int GL_MAXSEG = 256;
struct ListSrc {
    const uint32_t *glist, *gcnt;   // null: the list is dense (rlist)
    int n_gseg, rpb;
    uint32_t gen_grid;
    uint32_t *segfirst;             // [gen_grid] out: read index of the first entry of every block's range (the heavy pass anchors its window there)
    struct GenGeo *geo_out;         // out (block 0): the geometry, for the heavy pass
};
struct GenGeo {            // geometry of the general pass, decided on the device by k_gcompact
    uint32_t n_list;       // entries of the dense list
    uint32_t tpb;          // tiles per block of k_tile<LIST>
    uint32_t n_seg;        // its blocks that have tiles
    uint32_t live_counted; // 1: ctr[28] holds the entries not flagged GL_LONG (k_long's batches); the tile kernel leaves at once when there are none
};

// per-read state words kept in LDS for the chunk lanes
enum : int { S_OFF8, S_LOHI, S_FF, S_REV, S_ERR, S_CB2, S_WORDS };
enum : int { G_M, G_R0, G_RC, G_WORDS };   // per-segment words; G_RC = owner lane | chunk base << 8 (the read's offset is st[S_OFF8][lane])

struct WaveLds {
    uint16_t cigA[T_MAXOPS * TILE];   // len<<4|op in 16 bits: the tile path takes reads whose op lengths sum to < 4096
    uint16_t cigB[T_MAXOPS * TILE];   // scratch during trimming, chunk map during P2 / P4
    uint32_t st[S_WORDS * TILE];
    uint32_t seg[G_WORDS * T_SEGCAP];
};
struct BlockLds {
    uint32_t win[AMP_NSYM * T_W];
    uint32_t lut[16];                 // BAM base code -> byte offset of its count plane
    uint32_t dcount;                  // light entries (indels only) of this block's deferred-list segment, from its front
    uint32_t dcount2;                 // heavy entries (whole read / exact status), from its back
    uint32_t gpre[GL_MAXSEG + 1];     // LIST without k_gcompact: entries in front of every list segment
    WaveLds wv[T_WAVES];
};

// one CIGAR column in LDS (stride TILE words)
struct LdsCig {
    lds_u16 *p;
    __device__ __forceinline__ uint32_t get(int i) const { return p[i * TILE]; }
    __device__ __forceinline__ void set(int i, uint32_t v) const { p[i * TILE] = (uint16_t)v; }
};

// Lanes of one wave hand data to each other through LDS only: the fences are restricted to the local
// address space, so the wait is `s_waitcnt lgkmcnt(0)` and global loads / stores in flight (prefetched
// qualities, the outputs of P3) are not drained at every phase change.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

__device__ __forceinline__ void lds_add(lds_u32 *p, uint32_t v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct TileCtx {
    lds_u32 *win;
    int32_t win_base;
    uint32_t wlim;      // min(T_W, ref_len - win_base): window positions that are real
    uint32_t *counts;
    EventBuf eb;
    uint32_t G;
};

__device__ __forceinline__ void tile_add(const TileCtx &t, int32_t r, uint32_t col) {
    uint32_t d = (uint32_t)(r - t.win_base);
    if (d < (uint32_t)T_W) lds_add(t.win + col * T_W + d, 1u);
    else atomicAdd(&t.counts[(size_t)r * AMP_NSYM + col], 1u);
}

struct TileSink {
    const TileCtx &t;
    uint32_t read;
    __device__ void add(int32_t r, uint32_t col) { tile_add(t, r, col); }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) { t.eb.record(pos, read, lo, hi); }
};

// The skip-ahead variant of the exact walk for REGULAR reads: M/=/X runs and the end clips
// are stepped over in O(1) (their bases are counted by the chunk lanes / have no effect);
// deletions, reference skips and insertion runs take exactly the path of count_read_walk.
template <class CB, class Sink, class QF>
__device__ int count_regular_skip(const KParams &P, const CB &cig, int n, int32_t ref_start, int32_t lseq,
                                  int32_t qs, int32_t qe, const QF &qual, Sink &sink) {
    const int32_t ref_end = ref_start + reference_length(cig, n);
    const uint32_t G = (uint32_t)P.ref_len;
    const int32_t mq = P.min_quality;
    PairIter<CB> it;
    it.init(cig, n, ref_start);
    int32_t q, r;
    bool pending = false;
    int32_t pend_q = 0, pend_r = 0;
    for (;;) {
        if (pending) {
            q = pend_q; r = pend_r; pending = false;
        } else {
            while (it.j >= it.len) {   // step over whole ops that cannot produce effects here
                if (it.k + 1 >= it.n) return 0;
                uint32_t v = it.c.get(it.k + 1), op = v & 15u;
                int32_t len = (int32_t)(v >> 4);
                if (is_match_op(op)) { it.q += len; it.r += len; ++it.k; it.j = 0; it.len = 0; it.op = op; }
                else if (op == OP_S) { it.q += len; ++it.k; it.j = 0; it.len = 0; it.op = op; }
                else break;
            }
            if (!it.next(q, r)) break;
        }
        if (q < 0) {
            if ((uint32_t)r >= G) return AMP_RS_INDEX_REF;
            sink.add(r, 5u);
            continue;
        }
        if (r >= 0) {
            // a match base (handed back by an insertion scan, or first of an op reached through
            // a hard clip): the chunk lanes count it; skip the rest of its op
            if (is_match_op(it.op) && it.j < it.len) { int32_t rest = it.len - it.j; it.q += rest; it.r += rest; it.j = it.len; }
            continue;
        }
        if ((int32_t)qual(q) < mq) continue;
        if (q < qs) continue;
        if (q >= qe) break;
        const int32_t q0 = q;
        bool q_none = false;
        while (r < 0 && !q_none && q < qe) {
            if ((int32_t)qual(q) < mq) break;
            if (!it.next(q, r)) return AMP_RS_INDEX_PAIRS;
            if (q < 0) q_none = true;
        }
        int32_t lo, hi;
        if (r == 0) {
            if (q_none) return AMP_RS_TYPE;
            py_slice(q0, q + 1, lseq, lo, hi);
        } else if (q_none) {
            py_slice(q0 - 1, lseq, lseq, lo, hi);
        } else {
            py_slice(q0 - 1, q, lseq, lo, hi);
        }
        int32_t ins_pos;
        if (r < 0) ins_pos = ref_end;
        else { ins_pos = r; pending = true; pend_q = q; pend_r = r; }
        ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;
        if ((uint32_t)ins_pos >= G) return AMP_RS_INDEX_REF;
        sink.event(ins_pos, lo, hi);
    }
    return 0;
}

// qualities through an 8-byte register cache (reads start on 8-byte boundaries): the bases of an
// insertion cost one or two memory round trips instead of one each
struct QualAt {
    const uint8_t *q;
    mutable int32_t blk = -1;
    mutable uint32_t lo = 0, hi = 0;
    __device__ uint32_t operator()(int32_t k) const {
#ifdef AMP_ABL_QUALCONST
        return 40u;
#endif
        if ((k >> 3) != blk) {
            blk = k >> 3;
            const uint2 v = *(const uint2 *)(q + (int64_t)blk * 8);
            lo = v.x; hi = v.y;
        }
        return (((k & 4) ? hi : lo) >> ((k & 3) * 8)) & 0xFFu;
    }
};


// Sink of the in-tile indel walk: '-' counts go through the block's window, insertion events are staged in the
// wave's (by then idle) segment table and leave for the list with ONE reservation per tile (a returning global
// atomic per event serialises in L2).  Events beyond the staging capacity take the slow path.
constexpr uint32_t T_EVCAP = 144;     // = G_WORDS * T_SEGCAP / 4 events of four words
struct TileEvSink {
    const TileCtx &t;
    uint32_t read;
    lds_u32 *stage, *cursor;
    __device__ void add(int32_t r, uint32_t col) {
#ifndef AMP_ABL_NOADD
        tile_add(t, r, col);
#endif
    }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
#ifdef AMP_ABL_NOEVENT
        return;
#endif
        const uint32_t k = __hip_atomic_fetch_add(cursor, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k < T_EVCAP) {
            stage[k * 4] = (uint32_t)pos; stage[k * 4 + 1] = read; stage[k * 4 + 2] = (uint32_t)lo; stage[k * 4 + 3] = (uint32_t)hi;
            atomicAdd(&t.eb.ins_at[pos], 1u);
        } else {
            t.eb.record(pos, read, lo, hi);
        }
    }
};

// A(1) C(2) G(4) T(8) -> 0..3, N(15) -> 4, anything else -> 15
__device__ __forceinline__ uint32_t col_of_code(uint32_t code) {
    const uint32_t lo = 0xFFF2F10Fu;  // codes 0..7
    const uint32_t hi = 0x4FFFFFF3u;  // codes 8..15
    uint32_t x = (code & 8u) ? hi : lo;
    return (x >> ((code & 7u) * 4u)) & 15u;
}

// sum of the W bytes starting at byte `b` (0..7) of the 16-byte group w[0..3]
template <int W>
__device__ __forceinline__ uint32_t window_sum_at(const uint32_t (&w)[4], int b) {
    const int d = b >> 2, sh = (b & 3) * 8;
    uint32_t x = sh ? __builtin_amdgcn_alignbit(w[d + 1], w[d], sh) : w[d];
    if (W <= 4) {
        if (W < 4) x &= (1u << (W * 8)) - 1u;
        return __builtin_amdgcn_sad_u8(x, 0u, 0u);
    }
    uint32_t y = sh ? __builtin_amdgcn_alignbit(d + 2 < 4 ? w[d + 2] : 0u, w[d + 1], sh) : w[d + 1];
    if (W < 8) y &= (1u << ((W - 4) * 8)) - 1u;
    return __builtin_amdgcn_sad_u8(y, 0u, __builtin_amdgcn_sad_u8(x, 0u, 0u));
}

// bit b set: the W-byte window starting at byte b of w[] sums to less than thr
template <int W>
__device__ __forceinline__ uint32_t window_fail_bits(const uint32_t (&w)[4], uint32_t thr) {
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) m |= (window_sum_at<W>(w, b) < thr ? 1u : 0u) << b;
    return m;
}

__device__ __forceinline__ uint32_t window_fail_bits_dyn(int Wd, const uint32_t (&w)[4], uint32_t thr) {
    switch (Wd) {
        case 1: return window_fail_bits<1>(w, thr); case 2: return window_fail_bits<2>(w, thr);
        case 3: return window_fail_bits<3>(w, thr); case 4: return window_fail_bits<4>(w, thr);
        case 5: return window_fail_bits<5>(w, thr); case 6: return window_fail_bits<6>(w, thr);
        case 7: return window_fail_bits<7>(w, thr); default: return window_fail_bits<8>(w, thr);
    }
}

// exclusive prefix sum over the 64 lanes; total in every lane
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t &total) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    total = __shfl(incl, 63);
    return incl - v;
}

// Classification of a final CIGAR.  regular: H* S* (M|=|X|I|D|N)* S* H* with query length ==
// lseq.  nseg = number of match ops; plain = regular with no I/D/N at all.
template <class CB>
__device__ void classify(const CB &c, int n, int32_t lseq, bool &regular, bool &plain, int &nseg) {
    int phase = 0;  // 0 lead H, 1 lead S, 2 body, 3 trail S, 4 trail H
    int32_t q = 0;
    int nother = 0;
    regular = true; nseg = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t v = c.get(i), op = v & 15u;
        int32_t len = (int32_t)(v >> 4);
        if (op == OP_H) {
            if (phase == 0) continue;
            if (phase >= 2) { phase = 4; continue; }
            regular = false; break;   // H after a leading S
        } else if (op == OP_S) {
            if (phase <= 1) { phase = 1; q += len; }
            else if (phase <= 3) { phase = 3; q += len; }
            else { regular = false; break; }
        } else if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_D || op == OP_N) {
            if (phase > 2) { regular = false; break; }
            phase = 2;
            if (is_match_op(op)) { ++nseg; q += len; }
            else { ++nother; if (op == OP_I) q += len; }
        } else {
            regular = false; break;
        }
    }
    if (q != lseq) regular = false;
    plain = regular && nother == 0;
}

// ---- chunk phases -------------------------------------------------------------------------------
// Both chunk loops run as blocks of T_UNROLL chunks per lane: stage A issues every global load of
// the block, stage B consumes them.  FULL = every lane of the wave has T_UNROLL chunks (no bounds
// tests); the last, partial block of a round runs the checked variant.  Loads are unconditional:
// a chunk may read up to 8 bytes past its own 8 bytes, so `qual` and `seq` need 16 bytes of
// readable slack after the last read (documented in amplihip.h).

typedef short amp_short2 __attribute__((ext_vector_type(2)));

// bit b set: the W-byte window starting at byte b of the 16-byte group sums to less than thr
template <int W>
__device__ __forceinline__ uint32_t window_fail_bits16(const uint2 lo, const uint2 hi, uint32_t thr) {
    if (W == 4) {
        // four sliding 4-byte sums per instruction, packed as u16
        const uint64_t a = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)lo.x | ((uint64_t)lo.y << 32), 0u, 0ull);
        const uint64_t b = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)lo.y | ((uint64_t)hi.x << 32), 0u, 0ull);
        const uint32_t t2 = thr | (thr << 16);
        const uint32_t d[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            amp_short2 x = __builtin_bit_cast(amp_short2, d[k]) - __builtin_bit_cast(amp_short2, t2);   // negative: sum < thr
            const uint32_t u = __builtin_bit_cast(uint32_t, x);
            m |= ((u >> 15) & 1u) << (2 * k);
            m |= (u >> 31) << (2 * k + 1);
        }
        return m;
    }
    const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
    return window_fail_bits<W>(w, thr);
}

struct ChunkEnv {
    const lds_u8 *cmap;
    lds_u32 *st;
    lds_u32 *seg;
    lds_u32 *win;
    lds_u32 *lut;
    const uint8_t *qual;
    const uint8_t *seq;
    uint32_t *counts;
    int32_t win_base;
    uint32_t wlim, G;
    int32_t mq;
};

template <int W, bool FULL>
__device__ __forceinline__ void p2_block(const ChunkEnv &E, uint32_t cc, uint32_t lim, uint32_t base, uint32_t thr) {
    uint2 a0[T_UNROLL], a1[T_UNROLL];
    uint32_t rr[T_UNROLL];
    int32_t jj[T_UNROLL];
    // every address of the block first, then its loads back to back: a load issued between the address
    // computations of the next chunk gets waited for as soon as the compiler reuses one of its registers
    const uint8_t *qp[T_UNROLL];
#pragma unroll
    for (int u = 0; u < T_UNROLL; ++u) {
        const uint32_t c = cc + 64u * u;
        rr[u] = 0; jj[u] = 0; qp[u] = E.qual;
        if (FULL || c < lim) {
            const uint32_t r = E.cmap[c];
            const int32_t rlo = (int32_t)(E.st[S_LOHI * TILE + r] & 0xFFFFu);
            const int32_t j0 = ((int32_t)(c + base - E.st[S_CB2 * TILE + r]) + (rlo >> 3)) * 8;
            qp[u] = E.qual + (int64_t)E.st[S_OFF8 * TILE + r] * 8 + j0;
            rr[u] = r; jj[u] = j0;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < T_UNROLL; ++u) {
        a0[u] = make_uint2(0, 0); a1[u] = make_uint2(0, 0);
        if (FULL || cc + 64u * u < lim) {
            a0[u] = *(const uint2 *)qp[u];
            a1[u] = *(const uint2 *)(qp[u] + 8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < T_UNROLL; ++u) {
        if (FULL || cc + 64u * u < lim) {
            const uint32_t r = rr[u];
            const int32_t j0 = jj[u];
            const uint32_t lohi = E.st[S_LOHI * TILE + r];
            const int32_t rlo = (int32_t)(lohi & 0xFFFFu), rhi = (int32_t)(lohi >> 16);
            uint32_t fail = window_fail_bits16<W>(a0[u], a1[u], thr);
            int32_t blo = rlo - j0, bhi = rhi - W - j0;          // starts j0+b must lie in [rlo, rhi - W]
            blo = blo < 0 ? 0 : blo; bhi = bhi > 7 ? 7 : bhi;
            fail &= (0xFFu >> (7 - bhi)) & (0xFFu << blo);
            if (fail) {
                const bool rv = E.st[S_REV * TILE + r] != 0;
                const uint32_t vf = (uint32_t)(j0 + (__builtin_ffs((int)fail) - 1) - rlo);            // first failing window start
                const uint32_t vr = 0xFFFFu - (uint32_t)(j0 + (31 - __builtin_clz(fail)) + W - rlo);  // last failing window end
                __hip_atomic_fetch_min(E.st + S_FF * TILE + r, rv ? vr : vf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int W>
__device__ __forceinline__ void p2_round(const ChunkEnv &E, int lane, uint32_t lim, uint32_t base, uint32_t thr) {
    const uint32_t step = 64u * T_UNROLL;
    const uint32_t nfull = (lim / step) * step;
    uint32_t cc = (uint32_t)lane;
    for (; cc < nfull; cc += step) p2_block<W, true>(E, cc, lim, base, thr);
    if (nfull < lim) p2_block<W, false>(E, nfull + (uint32_t)lane, lim, base, thr);
}

template <bool FULL>
__device__ __forceinline__ void p4_block(const ChunkEnv &E, int lane, uint32_t cc, uint32_t lim, uint32_t base) {
    uint2 aq[T_UNROLL4];
    uint32_t as_[T_UNROLL4], sgs[T_UNROLL4];
    int32_t jj[T_UNROLL4];
    int64_t rbs[T_UNROLL4];
#pragma unroll
    for (int u = 0; u < T_UNROLL4; ++u) {      // addresses first, loads back to back (see p2_block)
        const uint32_t c = cc + 64u * u;
        sgs[u] = 0; jj[u] = 0; rbs[u] = 0;
        if (FULL || c < lim) {
            const uint32_t sg = E.cmap[c];
            const int32_t m0 = (int32_t)(E.seg[G_M * T_SEGCAP + sg] & 0xFFFFu);
            const int32_t j0 = ((int32_t)(c + base - (E.seg[G_RC * T_SEGCAP + sg] >> 8)) + (m0 >> 3)) * 8;
            rbs[u] = (int64_t)E.st[S_OFF8 * TILE + (E.seg[G_RC * T_SEGCAP + sg] & 0xFFu)] * 8 + j0;
            sgs[u] = sg; jj[u] = j0;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < T_UNROLL4; ++u) {
        aq[u] = make_uint2(0, 0); as_[u] = 0;
        if (FULL || cc + 64u * u < lim) {
            aq[u] = *(const uint2 *)(E.qual + rbs[u]);
            as_[u] = *(const uint32_t *)(E.seq + (rbs[u] >> 1));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t rot = ((uint32_t)lane >> 2) & 7u;
#pragma unroll
    for (int u = 0; u < T_UNROLL4; ++u) {
        if (FULL || cc + 64u * u < lim) {
            const uint32_t sg = sgs[u];
            const uint32_t mm = E.seg[G_M * T_SEGCAP + sg];
            const int32_t m0 = (int32_t)(mm & 0xFFFFu), m1 = (int32_t)(mm >> 16);
            const int32_t j0 = jj[u];
            const int32_t d0 = (int32_t)E.seg[G_R0 * T_SEGCAP + sg] + (j0 - m0) - E.win_base;   // window offset of base 0
            const uint2 qw = aq[u];
            uint32_t sw = as_[u];
            // per-byte flags (bit 7): base inside [m0, m1) and quality >= min_quality
            int32_t klo = m0 - j0, khi = m1 - j0;
            klo = klo < 0 ? 0 : klo; khi = khi > 8 ? 8 : khi;
            const uint64_t inm = (khi >= 8 ? ~0ull : ((1ull << (khi * 8)) - 1ull)) & ~((1ull << (klo * 8)) - 1ull);
            uint32_t ok0, ok1;
            if (E.mq <= 128) {
                const uint32_t mqb = (uint32_t)E.mq * 0x01010101u;
                ok0 = ((((qw.x & 0x7F7F7F7Fu) | 0x80808080u) - mqb) | qw.x) & 0x80808080u;
                ok1 = ((((qw.y & 0x7F7F7F7Fu) | 0x80808080u) - mqb) | qw.y) & 0x80808080u;
            } else {
                ok0 = ok1 = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    ok0 |= (((qw.x >> (8 * b)) & 0xFFu) >= (uint32_t)E.mq ? 0x80u : 0u) << (8 * b);
                    ok1 |= (((qw.y >> (8 * b)) & 0xFFu) >= (uint32_t)E.mq ? 0x80u : 0u) << (8 * b);
                }
            }
            ok0 &= (uint32_t)inm; ok1 &= (uint32_t)(inm >> 32);
            // a counted base with a code outside A C G T N, or a chunk that leaves the LDS window: careful path
            sw = ((sw & 0x0F0F0F0Fu) << 4) | ((sw >> 4) & 0x0F0F0F0Fu);   // base k at bits [4k, 4k+4)
            uint32_t pc = sw - ((sw >> 1) & 0x55555555u);
            pc = (pc & 0x33333333u) + ((pc >> 2) & 0x33333333u);         // per-nibble popcount (0..4)
            const uint32_t good = (pc ^ (pc >> 2)) & ~(pc >> 1) & 0x11111111u;   // popcount 1 (A C G T) or 4 (N)
            const uint32_t nm = (khi >= 8 ? 0xFFFFFFFFu : ((1u << (khi * 4)) - 1u)) & ~((1u << (klo * 4)) - 1u);
            const bool safe = (~good & nm & 0x11111111u) == 0u && (uint32_t)d0 <= E.wlim - 8u && E.wlim >= 8u;
            if (safe) {
                // rotate the 8 bases by `rot` so that lanes serviced together spread over the banks
                const uint32_t oa = (rot & 4u) ? ok1 : ok0, ob = (rot & 4u) ? ok0 : ok1;
                const uint32_t sh = (rot & 3u) * 8u;
                const uint32_t k0 = __builtin_amdgcn_alignbit(ob, oa, sh), k1 = __builtin_amdgcn_alignbit(oa, ob, sh);
                const uint32_t sr = __builtin_amdgcn_alignbit(sw, sw, rot * 4u);
                lds_u8 *const wbase = (lds_u8 *)E.win + (uint32_t)d0 * 4u;
                uint32_t plane[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) {    // all eight table reads first: they cannot move past the atomics
                    const uint32_t code4 = b == 0 ? (sr << 2) & 0x3Cu : (sr >> (4 * b - 2)) & 0x3Cu;
                    plane[b] = *(lds_u32 *)((lds_u8 *)E.lut + code4);
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const uint32_t val = ((b < 4 ? k0 : k1) >> ((b & 3) * 8 + 7)) & 1u;
                    const uint32_t bb4 = ((rot + (uint32_t)b) & 7u) * 4u;
                    lds_add((lds_u32 *)(wbase + plane[b] + bb4), val);
                }
            } else {
                bool bad = false;
#pragma unroll 1
                for (int b = 0; b < 8; ++b) {
                    const uint32_t okb = ((b < 4 ? ok0 : ok1) >> ((b & 3) * 8 + 7)) & 1u;
                    if (!okb) continue;
                    const uint32_t col = col_of_code((sw >> (4 * b)) & 15u);
                    const int32_t rp = E.win_base + d0 + b;
                    const uint32_t d = (uint32_t)(d0 + b);
                    if (col > 4u || (uint32_t)rp >= E.G) bad = true;
                    else if (d < (uint32_t)T_W) lds_add(E.win + col * T_W + d, 1u);
                    else atomicAdd(&E.counts[(size_t)rp * AMP_NSYM + col], 1u);
                }
                if (bad) E.st[S_ERR * TILE + (E.seg[G_RC * T_SEGCAP + sg] & 0xFFu)] = 1u;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void p4_round(const ChunkEnv &E, int lane, uint32_t lim, uint32_t base) {
    const uint32_t step = 64u * T_UNROLL4;
    const uint32_t nfull = (lim / step) * step;
    uint32_t cc = (uint32_t)lane;
    for (; cc < nfull; cc += step) p4_block<true>(E, lane, cc, lim, base);
    if (nfull < lim) p4_block<false>(E, lane, nfull + (uint32_t)lane, lim, base);
}

// ---------------------------------------------------------------------------------------
// Per-read hand-over between the kernels of the split pipeline (variant 3): what P1 knows about a read
// after the primer clips.  The primer-trimmed CIGAR itself travels through the read's output slot.
struct SplitDesc {
    int32_t *pos;      // reference_start after the start clip
    uint32_t *lohi;    // aligned-quality window lo | hi << 16
    uint32_t *meta;    // n ops | trim flags << 8 | status << 16 | SD_* bits
    uint32_t *ff;      // result of the window scan (S_FF encoding), 0xFFFF = no full window failed
};
constexpr uint32_t SD_DEFER = 1u << 24, SD_HAVE_QUAL = 1u << 25, SD_CAN_Q = 1u << 26, SD_REV = 1u << 27;

// LIST: the tile's reads are entries [tile * 64, tile * 64 + 64) of `rlist` (ascending read order inside each
// fast-kernel block's share), the number of entries and the tiles per block come from `geo` (device memory).
// dcnt_stride = blocks the list counts were sized for (the grid; LIST: the host's worst-case grid).
#ifdef AMP_DEV
#define AMP_PHASES_PARAM , uint32_t phases
#define AMP_PHASES_ARG(x) , (x)
#else
#define AMP_PHASES_PARAM
#define AMP_PHASES_ARG(x)
#endif
template <bool STAMPS, bool SPLIT, bool LIST>
__global__ void __launch_bounds__(T_WAVES * 64, 4)
k_tile(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *counts, EventBuf eb, uint32_t *dlist,
       uint32_t *dcnt, int tiles_per_block, SplitDesc sd, const uint32_t *rlist, const GenGeo *geo,
       uint32_t dcnt_stride, ListSrc ls AMP_PHASES_PARAM) {
#ifndef AMP_DEV
    constexpr uint32_t phases = 0xFFu;     // the shipped library cannot mask phases off
#endif
    __shared__ BlockLds L;
    unsigned long long *const ctr = eb.ctr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool direct = LIST && ls.glist != nullptr;
    uint32_t n_list_direct = 0;
    if (direct && (uint32_t)ctr[29] != P.epoch) {
        // no block of this launch's fast kernel handed a read over: an empty pass (what the code below finds out after adding up the
        // list lengths of all blocks; the batches of an amplicon run are like this nine times in ten)
        if (blockIdx.x == 0 && tid == 0) {
            ls.geo_out->n_list = 0u; ls.geo_out->tpb = (uint32_t)T_WAVES; ls.geo_out->n_seg = 0u; ls.geo_out->live_counted = 0u;
            ctr[7] = 0ull;
        }
        if (tid == 0) { dcnt[blockIdx.x] = 0; dcnt[5 * dcnt_stride + 64 + blockIdx.x] = 0; }
        return;
    }
    if (direct) {
        if (tid <= GL_MAXSEG) L.gpre[tid] = tid < ls.n_gseg ? ls.gcnt[tid] : 0u;
        __syncthreads();
        if (wave == 0) {
            const uint32_t a0 = L.gpre[4 * lane], a1 = L.gpre[4 * lane + 1], a2 = L.gpre[4 * lane + 2], a3 = L.gpre[4 * lane + 3];
            uint32_t total;
            const uint32_t ex = wave_excl_scan(a0 + a1 + a2 + a3, lane, total);
            L.gpre[4 * lane] = ex; L.gpre[4 * lane + 1] = ex + a0; L.gpre[4 * lane + 2] = ex + a0 + a1; L.gpre[4 * lane + 3] = ex + a0 + a1 + a2;
            if (lane == 0) L.gpre[GL_MAXSEG] = total;
        }
        __syncthreads();
        n_list_direct = L.gpre[GL_MAXSEG];
        const uint32_t tiles = (n_list_direct + TILE - 1) / TILE;
        uint32_t tpb = (tiles + ls.gen_grid - 1) / ls.gen_grid;
        tpb = ((tpb + T_WAVES - 1) / T_WAVES) * T_WAVES;
        if (tpb < (uint32_t)T_WAVES) tpb = T_WAVES;
        tiles_per_block = (int)tpb;
        if (blockIdx.x == 0 && tid == 0) {
            ls.geo_out->n_list = n_list_direct; ls.geo_out->tpb = tpb; ls.geo_out->n_seg = (tiles + tpb - 1) / tpb; ls.geo_out->live_counted = 0u;
            ctr[7] = n_list_direct;                   // (amp_debug_counters: reads of the last batch that took the general pass)
        }
    } else if (LIST) tiles_per_block = (int)geo->tpb;
    const int64_t n = direct ? (int64_t)n_list_direct : LIST ? (int64_t)geo->n_list : rd.n_reads;
    // entry li of the list
    const auto list_entry = [&](int64_t li) -> uint32_t {
        if (!direct) return rlist[li];
        int lo = 0, hi = ls.n_gseg - 1;               // the last segment that starts at or in front of li
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int64_t)L.gpre[mid] <= li) lo = mid; else hi = mid - 1; }
        return ls.glist[(size_t)lo * (size_t)ls.rpb + (size_t)(li - (int64_t)L.gpre[lo])];
    };
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_block;
    const int64_t tile_end = tile_begin + tiles_per_block < n_tiles ? tile_begin + tiles_per_block : n_tiles;
    if (tile_begin >= tile_end || (LIST && !direct && geo->live_counted && ctr[28] == 0ull)) {
        if (threadIdx.x == 0) { dcnt[blockIdx.x] = 0; dcnt[5 * dcnt_stride + 64 + blockIdx.x] = 0; }
        return;
    }

    if (LIST && ls.segfirst && tid == 0) ls.segfirst[blockIdx.x] = list_entry(tile_begin * TILE) & GL_INDEX_MASK;
    lds_u32 *const win = (lds_u32 *)L.win;
    lds_u32 *const lut = (lds_u32 *)L.lut;
    for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) win[i] = 0;
    if (tid == 0) { L.dcount = 0; L.dcount2 = 0; }
    if (tid < 16) { uint32_t c = col_of_code((uint32_t)tid); lut[tid] = c <= 4u ? c * (uint32_t)(T_W * 4) : 0u; }
    int32_t win_base = NO_WINDOW;
    lds_u32 *const st = (lds_u32 *)L.wv[wave].st;
    lds_u32 *const seg = (lds_u32 *)L.wv[wave].seg;
    lds_u16 *const cigA = (lds_u16 *)L.wv[wave].cigA;
    lds_u16 *const cigB = (lds_u16 *)L.wv[wave].cigB;
    lds_u8 *const cmap = (lds_u8 *)L.wv[wave].cigB;
    TileCtx tc{win, 0, 0u, counts, eb, (uint32_t)P.ref_len};
    const int32_t mq = P.min_quality;
    const int32_t Wd = P.window;
    const uint32_t mqc = (uint32_t)(mq > 256 ? 256 : mq);          // sums of W bytes never reach 256*W
    unsigned long long n_err = 0;
    constexpr bool stamps = STAMPS;
    const unsigned long long t_kernel0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    uint32_t bs_rebase = 0, bs_c2 = 0, bs_c4 = 0;

    for (int64_t t0 = tile_begin; t0 < tile_end; t0 += T_WAVES) {
        // ---- window management (uniform over the block) ------------------------------------
        const int32_t first_pos = rd.pos[LIST ? (int64_t)(list_entry(t0 * TILE) & GL_INDEX_MASK) : t0 * TILE];
        if (win_base == NO_WINDOW || first_pos < win_base || first_pos - win_base >= T_W / 2) {
            __syncthreads();
            if (win_base != NO_WINDOW) {
                for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) {
                    uint32_t v = win[i];
                    if (v) {
                        int sym = i / T_W, d = i - sym * T_W;
                        atomicAdd(&counts[(size_t)(win_base + d) * AMP_NSYM + sym], v);
                        win[i] = 0;
                    }
                }
            }
            win_base = first_pos & ~31;
            ++bs_rebase;
            __syncthreads();
        }
        tc.win_base = win_base;
        {
            int64_t lim = (int64_t)P.ref_len - win_base;
            tc.wlim = lim <= 0 ? 0u : (lim > T_W ? (uint32_t)T_W : (uint32_t)lim);
        }
        const ChunkEnv env{cmap, st, seg, win, lut, rd.qual, rd.seq, counts, win_base, tc.wlim, (uint32_t)P.ref_len, mq};
        const int64_t tile = t0 + wave;
        if (tile >= tile_end) continue;

        unsigned long long tprev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
#define AMP_STAMP(k) do { if (stamps) { unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[k] += tn - tprev; tprev = tn; } } while (0)
        // =================================== P1: lane = read ===================================
        const int64_t li = tile * TILE + lane;           // row of the batch, or of the list
        bool valid = li < n;
        uint32_t lent = 0;
        if (LIST && valid) { lent = list_entry(li); if (lent & GL_LONG) valid = false; }
        if (LIST && !__ballot(valid)) continue;            // a tile of reads that k_long took
        const int64_t i = LIST ? (int64_t)(lent & GL_INDEX_MASK) : li;
        const bool status_wanted = LIST && (lent & GL_STATUS_ONLY);      // the fast kernel counted it: exact status only
        int32_t lseq = 0, pos = 0, tlen = 0;
        uint32_t flag = 0, c0 = 0, off8 = 0, meta = 0, lohi = 0;
        int ncig = 0;
        if (valid) {
            lseq = (int32_t)rd.lseq[i]; c0 = rd.cig_off32[i]; off8 = rd.seq_off8[i];
            if (SPLIT) { meta = sd.meta[i]; lohi = sd.lohi[i]; pos = sd.pos[i]; }
            else { pos = rd.pos[i]; flag = rd.flag[i]; tlen = rd.tlen[i]; ncig = (int)(rd.cig_off32[i + 1] - c0); }
        }
        const size_t slot = (size_t)c0 + 3 * (size_t)(valid ? i : 0);
        const int64_t boff = (int64_t)off8 * 8;
        const uint8_t *qual = rd.qual + boff;
        bool defer_full = SPLIT ? (meta & SD_DEFER) != 0 : valid && (ncig + 3 > T_MAXOPS || (uint32_t)lseq >= 65536u || status_wanted);
        LdsCig cur{cigA + lane}, tmp{cigB + lane};
        uint32_t q0 = 0xFFu;
        if (!SPLIT) {
            // second level of loads, all issued before any is waited for: the first quality byte (QUAL '*' marker)
            // and every CIGAR word of the tile (a loop over k with a load and a wait per turn made this
            // several dependent round trips)
            const bool ld = valid && !defer_full;
            int maxn = ld ? ncig : 0;
            for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(maxn, o); maxn = t > maxn ? t : maxn; }
            uint32_t w[T_MAXOPS - 3];
#pragma unroll
            for (int k = 0; k < T_MAXOPS - 3; ++k) { w[k] = 0; if (k < maxn && ld && k < ncig) w[k] = rd.cig[c0 + k]; }
            if (ld && lseq > 0) q0 = qual[0];
            __builtin_amdgcn_sched_barrier(0);
            // the columns hold len<<4|op in 16 bits: a read takes the tile path when no length a trim could
            // produce (ops merge, clips add up) reaches 4096, i.e. when all its lengths sum to less
            uint32_t sumlen = 0;
#pragma unroll
            for (int k = 0; k < T_MAXOPS - 3; ++k)
                if (k < maxn && ld && k < ncig) { cur.set(k, w[k]); sumlen += w[k] >> 4; }
            if (sumlen >= 4096u) defer_full = true;
        }
        const bool mine = valid && !defer_full;
        const bool have_qual = SPLIT ? (meta & SD_HAVE_QUAL) != 0 : mine && lseq > 0 && q0 != 0xFFu;
        const bool rev = SPLIT ? (meta & SD_REV) != 0 : (flag & 0x10u) != 0;
        TrimState ts{pos, ncig, 0u, 0};
        int32_t qs = 0, lo = 0, qlen = 0;
        bool can_q = false;
        if (SPLIT) {
            // k_trim did the primer clips: its CIGAR is in the read's output slot
            ts.n = (int)(meta & 0xFFu); ts.flags = (meta >> 8) & 0xFFu; ts.err = (int)((meta >> 16) & 0xFFu);
            can_q = (meta & SD_CAN_Q) != 0;
            qs = lo = (int32_t)(lohi & 0xFFFFu); qlen = (int32_t)(lohi >> 16) - lo;
            if (mine && !ts.err)
                for (int k = 0; k < ts.n; ++k) cur.set(k, out.new_cig[slot + k]);
        } else if (mine) {
            if (P.do_trim) {
                trim_primers(P, ts, flag, tlen, lseq, cur, tmp);
                if (!ts.err) can_q = quality_window(ts, lseq, have_qual, cur, qs, lo, qlen);
                if (cur.p != cigA + lane) {           // keep the CIGAR in buffer A: B becomes the chunk map
                    for (int k = 0; k < ts.n; ++k) cigA[lane + k * TILE] = (uint16_t)cur.get(k);
                    cur.p = cigA + lane; tmp.p = cigB + lane;
                }
            }
        }
        // the 3' end's shrinking windows (A:575-576, A:637-638) need at most W-1 bytes; fetch them now so that
        // they have landed when P3 wants them
        uint2 tw0 = make_uint2(0, 0), tw1 = make_uint2(0, 0);
        int32_t tab = 0;
        if (mine && can_q) {
            const int32_t first = (rev || qlen < Wd) ? lo : lo + qlen - Wd + 1;
            tab = first & ~7;
            tw0 = *(const uint2 *)(qual + tab);
            tw1 = *(const uint2 *)(qual + tab + 8);
        }
        // full windows start at aligned-quality indices [0, qlen - W]; chunks are 8 starts wide
        const bool par_scan = mine && can_q && Wd <= 8 && (phases & 2u);
        const int32_t hi = lo + qlen;
        uint32_t nch2 = 0;
        if (!SPLIT && par_scan && qlen >= Wd) nch2 = (uint32_t)(((hi - Wd) >> 3) - (lo >> 3) + 1);
        uint32_t total2;
        const uint32_t cb2 = wave_excl_scan(nch2, lane, total2);
        st[S_OFF8 * TILE + lane] = off8;
        st[S_LOHI * TILE + lane] = (uint32_t)lo | ((uint32_t)hi << 16);
        st[S_FF * TILE + lane] = (SPLIT && par_scan) ? sd.ff[i] : 0xFFFFu;
        st[S_REV * TILE + lane] = rev ? 1u : 0u;
        st[S_ERR * TILE + lane] = 0u;
        st[S_CB2 * TILE + lane] = cb2;

        AMP_STAMP(0);
        bs_c2 += total2;
        // =================================== P2: lane = chunk ===================================
        if (!SPLIT)
        for (uint32_t base = 0; base < total2; base += T_MAPCAP) {
            wave_sync();
            {   // read lanes publish chunk -> read for this round
                uint32_t a = cb2 > base ? cb2 : base, b = cb2 + nch2 < base + T_MAPCAP ? cb2 + nch2 : base + T_MAPCAP;
                for (uint32_t c = a; c < b; ++c) cmap[c - base] = (uint8_t)lane;
            }
            wave_sync();
            const uint32_t lim = total2 - base < (uint32_t)T_MAPCAP ? total2 - base : (uint32_t)T_MAPCAP;
            const uint32_t thr = mqc * (uint32_t)Wd;
            switch (Wd) {
                case 1: p2_round<1>(env, lane, lim, base, thr); break; case 2: p2_round<2>(env, lane, lim, base, thr); break;
                case 3: p2_round<3>(env, lane, lim, base, thr); break; case 4: p2_round<4>(env, lane, lim, base, thr); break;
                case 5: p2_round<5>(env, lane, lim, base, thr); break; case 6: p2_round<6>(env, lane, lim, base, thr); break;
                case 7: p2_round<7>(env, lane, lim, base, thr); break; default: p2_round<8>(env, lane, lim, base, thr); break;
            }
        }
        wave_sync();
        AMP_STAMP(1);

        // =================================== P3: lane = read ===================================
        int nseg = 0;
        bool counted = false;
        if (mine && !ts.err && P.do_trim && can_q) {
            int32_t iq;
            if (par_scan) {
                const uint32_t v = st[S_FF * TILE + lane];
                if (v != 0xFFFFu) {
                    iq = rev ? (int32_t)(0xFFFFu - v) : (int32_t)v;
                } else {
                    // no full window failed: the shrinking windows at the 3' end decide (A:575-576, A:637-638)
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
            } else {
                iq = quality_scan(qual + lo, qlen, Wd, mq, rev);
            }
            trim_quality_apply(ts, rev, iq, qlen, qs, cur, tmp);
            if (!ts.err && cur.p != cigA + lane) {
                for (int k = 0; k < ts.n; ++k) cigA[lane + k * TILE] = (uint16_t)cur.get(k);
                cur.p = cigA + lane; tmp.p = cigB + lane;
            }
        }
        if (mine) {
            int32_t reflen = 0;
            if (!ts.err) {
                uint32_t *home = out.new_cig + slot;
                for (int k = 0; k < ts.n; ++k) home[k] = cur.get(k);
                reflen = reference_length(cur, ts.n);
            }
            if (out.new_pos) out.new_pos[i] = ts.pos;
            if (out.new_ncig) out.new_ncig[i] = ts.err ? 0u : (uint32_t)ts.n;
            if (out.ref_len) out.ref_len[i] = ts.err ? 0 : reflen;
            if (out.trim_flags) out.trim_flags[i] = ts.err ? (uint8_t)0 : (uint8_t)ts.flags;
        }
        bool plain = false;
        if (mine && !ts.err && P.do_count) {
            bool regular;
            classify(cur, ts.n, lseq, regular, plain, nseg);
            if (!have_qual) regular = false;
            int e1 = 0, e2 = 0;
            if (regular) { (void)query_alignment_start(cur, ts.n, lseq, e1); (void)query_alignment_end(cur, ts.n, lseq, e2); }
            if (!regular || e1 || e2) { defer_full = true; nseg = 0; }
            else counted = true;
        }
        // segments: one per match op, allotted by a wave scan.  A tile whose reads have more match ops than the table holds
        // is counted in several passes (a read that did not fit used to go to the second pass, whole: on a list of
        // indel-heavy reads that was one read in twenty-five, and the second pass took as long as this kernel)
        if (counted && nseg > T_SEGCAP) { counted = false; defer_full = true; nseg = 0; }      // (one read alone overflows it)
#ifdef AMP_ABL_NOWALK
        const bool defer_indels = false;
#else
        const bool defer_indels = counted && !plain;   // deletions and insertion events: the in-tile walk below
#endif
        bool seg_todo = counted;
        for (;;) {
            uint32_t total_seg;
            const uint32_t sb = wave_excl_scan(seg_todo ? (uint32_t)nseg : 0u, lane, total_seg);
            const bool now = seg_todo && sb + (uint32_t)nseg <= (uint32_t)T_SEGCAP;            // (the first lane left always fits)
            uint32_t nch4 = 0;
            if (now) {
                int32_t q = 0, r = ts.pos;
                uint32_t sidx = sb;
                for (int k = 0; k < ts.n; ++k) {
                    uint32_t v = cur.get(k), op = v & 15u;
                    int32_t len = (int32_t)(v >> 4);
                    if (is_match_op(op)) {
                        if (len > 0) {
                            seg[G_M * T_SEGCAP + sidx] = (uint32_t)q | ((uint32_t)(q + len) << 16);
                            seg[G_R0 * T_SEGCAP + sidx] = (uint32_t)r;
                            seg[G_RC * T_SEGCAP + sidx] = (uint32_t)lane;
                            nch4 += (uint32_t)(((q + len + 7) >> 3) - (q >> 3));
                        } else {
                            seg[G_M * T_SEGCAP + sidx] = 0u; seg[G_R0 * T_SEGCAP + sidx] = 0u;
                            seg[G_RC * T_SEGCAP + sidx] = (uint32_t)lane;
                        }
                        ++sidx;
                        q += len; r += len;
                    } else if (op == OP_I || op == OP_S) q += len;
                    else if (op == OP_D || op == OP_N) r += len;
                }
            }
            if (!(phases & 4u)) nch4 = 0;
            uint32_t total4;
            const uint32_t cb4 = wave_excl_scan(nch4, lane, total4);

            AMP_STAMP(2);
            bs_c4 += total4;
            // =================================== P4: lane = chunk ===================================
            for (uint32_t base = 0; base < total4; base += T_MAPCAP) {
                wave_sync();
                if (now) {   // read lanes publish chunk -> segment for this round
                    uint32_t cpos = cb4;
                    for (int sgi = 0; sgi < nseg; ++sgi) {
                        const uint32_t mm = seg[G_M * T_SEGCAP + sb + sgi];
                        const uint32_t m0 = mm & 0xFFFFu, m1 = mm >> 16;
                        const uint32_t nc = m1 > m0 ? ((m1 + 7) >> 3) - (m0 >> 3) : 0u;
                        seg[G_RC * T_SEGCAP + sb + sgi] = (uint32_t)lane | (cpos << 8);
                        uint32_t a = cpos > base ? cpos : base, b = cpos + nc < base + T_MAPCAP ? cpos + nc : base + T_MAPCAP;
                        for (uint32_t c = a; c < b; ++c) cmap[c - base] = (uint8_t)(sb + sgi);
                        cpos += nc;
                    }
                }
                wave_sync();
                const uint32_t lim = total4 - base < (uint32_t)T_MAPCAP ? total4 - base : (uint32_t)T_MAPCAP;
                p4_round(env, lane, lim, base);
            }
            wave_sync();
            AMP_STAMP(3);
            seg_todo = seg_todo && !now;
            if (!__ballot(seg_todo)) break;
        }

        // ---- deletions, reference skips and insertion events of regular reads (A:714-715, A:730-748), lane = read:
        // the skip-ahead walk over the final CIGAR, which is still in the lane's LDS column; '-' goes through the
        // window, events through the idle segment table and out with one reservation for the whole tile --------
        bool indel_err = false;
        if (__ballot(defer_indels)) {
            lds_u32 *const evcur = st + S_CB2 * TILE;                 // (the chunk bases are no longer needed)
            if (lane == 0) *evcur = 0u;
            wave_sync();
            if (defer_indels) {
                TileEvSink sink{tc, (uint32_t)(read_base + (uint64_t)i), seg, evcur};
                int e1 = 0, e2 = 0;
                const int32_t qs2 = query_alignment_start(cur, ts.n, lseq, e1), qe2 = query_alignment_end(cur, ts.n, lseq, e2);
                indel_err = count_regular_ops(P, cur, ts.n, ts.pos, ts.pos + reference_length(cur, ts.n), lseq, qs2, qe2, QualAt{qual}, sink) != 0;   // exact status: second pass
            }
            wave_sync();
            const uint32_t nev_all = *evcur;
            const uint32_t nev = nev_all < T_EVCAP ? nev_all : T_EVCAP;
            if (nev) {
                const unsigned shard = blockIdx.x & (EV_SHARDS - 1);
                unsigned long long eb0 = 0;
                if (lane == 0) eb0 = atomicAdd(&eb.ctr[16 + shard], (unsigned long long)nev);
                eb0 = __shfl(eb0, 0);
                for (uint32_t k = (uint32_t)lane; k < nev; k += 64u)
                    if ((long long)(eb0 + k) < eb.cap)
                        eb.ev[(size_t)shard * (size_t)eb.cap + eb0 + k] =
                            amp_ins_event{(int32_t)seg[k * 4], seg[k * 4 + 1], (int32_t)seg[k * 4 + 2], (int32_t)seg[k * 4 + 3]};
            }
            wave_sync();
        }

        // ---- status / deferral (lane = read): one list reservation per wave -------------------------
        {
            uint32_t status = (uint32_t)ts.err;
            uint32_t entry = 0;
            bool has = false;
            if (valid) {
                if (status_wanted) { entry = (uint32_t)i | DEFER_STATUS_ONLY; has = true; status = 0; }
                else if (defer_full) { entry = (uint32_t)i; has = true; status = 0; }
                else if (!status && P.do_count) {
                    if (st[S_ERR * TILE + lane] || indel_err) { entry = (uint32_t)i | DEFER_STATUS_ONLY; has = true; }
                }
                if (status) ++n_err;
                if (out.status && !status_wanted) out.status[i] = (uint8_t)status;
            }
            // A single hot counter in global memory would serialise the whole chip (one returning atomic
            // per tile); every block appends to its OWN segment of the list through an LDS counter.
            // Light entries (only deletions / insertion events left to do) fill the segment from the front,
            // heavy ones (the whole read, or its exact status) from the back: two second-pass kernels.
            const bool light = has && entry == ((uint32_t)i | DEFER_INDELS);
            const unsigned long long m = __ballot(light), m2 = __ballot(has && !light);
            if (m) {
                uint32_t dbase = 0;
                if (lane == 0) dbase = __hip_atomic_fetch_add((lds_u32 *)&L.dcount, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                dbase = __shfl(dbase, 0);
                if (light) dlist[(size_t)tile_begin * TILE + dbase + __popcll(m & ((1ull << lane) - 1ull))] = entry;
            }
            if (m2) {
                uint32_t dbase = 0;
                if (lane == 0) dbase = __hip_atomic_fetch_add((lds_u32 *)&L.dcount2, (uint32_t)__popcll(m2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                dbase = __shfl(dbase, 0);
                if (has && !light)
                    dlist[(size_t)(tile_begin + tiles_per_block) * TILE - 1 - (dbase + __popcll(m2 & ((1ull << lane) - 1ull)))] = entry;
            }
        }
        AMP_STAMP(4);
        tacc[5] += 1;
    }

    // ---- final flush ------------------------------------------------------------------------
    const unsigned long long t_loopend = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    __syncthreads();
    const unsigned long long t_barrier = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (win_base != NO_WINDOW) {
        for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) {
            uint32_t v = win[i];
            if (v) {
                int sym = i / T_W, d = i - sym * T_W;
                atomicAdd(&counts[(size_t)(win_base + d) * AMP_NSYM + sym], v);
            }
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
    if (tid == 0) {
        dcnt[blockIdx.x] = L.dcount; dcnt[5 * dcnt_stride + 64 + blockIdx.x] = L.dcount2;
        if (L.dcount2) atomicOr(&ctr[24], 1ull);        // tells the heavy pass that it has something to do at all (sticky until amp_reset)
        // (no per-block atomic on a shared counter here: thousands of blocks on one address serialise;
        //  amp_debug_counters sums the per-block list counts instead)
    }
    if (stamps && lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&ctr[8 + k], tacc[k]);
        atomicAdd(&ctr[6], t_loopend - t_kernel0);      // time a wave spends in its tile loop
        atomicAdd(&ctr[7], t_barrier - t_loopend);      // ... waiting for the slowest wave of its block
        atomicAdd(&ctr[15], 1ull);
        if (tid == 0) {
            const unsigned long long dur = __builtin_amdgcn_s_memtime() - t_kernel0;
            atomicMax(&ctr[14], (dur << 16) | (unsigned long long)(blockIdx.x & 0xFFFF));            // slowest block and its id
            atomicMax(&ctr[5], ((0xFFFFFFFFFFFFull - dur) << 16) | (unsigned long long)(blockIdx.x & 0xFFFF));   // fastest block
            atomicAdd(&ctr[4], dur);
            dcnt[dcnt_stride + 64 + blockIdx.x * 4 + 0] = (uint32_t)dur; dcnt[dcnt_stride + 64 + blockIdx.x * 4 + 1] = bs_rebase;
        }
        if (lane == 0) { atomicAdd(&dcnt[dcnt_stride + 64 + blockIdx.x * 4 + 2], bs_c2); atomicAdd(&dcnt[dcnt_stride + 64 + blockIdx.x * 4 + 3], bs_c4); }
    }
}

// Geometry shared by the tile kernel and the second pass: block b owns tiles [b*tpb, (b+1)*tpb).
struct TileGrid { int64_t grid, tpb; };
static inline TileGrid tile_grid(int64_t n_reads, int n_cu) {
    const int64_t n_tiles = (n_reads + TILE - 1) / TILE;
    // 32 blocks per CU: two are resident, the rest are handed out as CUs free up, which evens out the
    // (measured) speed differences between blocks and XCDs.  Measured on 19.9 M reads: 8 blocks per CU
    // 3.45 ms, 16: 3.20, 32: 3.15, 64: 3.13, one tile per wave (the minimum): 3.34.
    int64_t tpb = (n_tiles + 32 * (int64_t)n_cu - 1) / (32 * (int64_t)n_cu);
    tpb = ((tpb + T_WAVES - 1) / T_WAVES) * T_WAVES;   // whole super-tiles per block
    if (tpb < T_WAVES) tpb = T_WAVES;
    return TileGrid{(n_tiles + tpb - 1) / tpb, tpb};
}

static inline int tile_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out,
                              uint32_t *counts, const EventBuf &eb, uint32_t *dlist, uint32_t *dcnt, int n_cu,
                              uint32_t phases, hipStream_t stream) {
    if (rd.n_reads == 0) return 0;
    const TileGrid tg = tile_grid(rd.n_reads, n_cu);
    const SplitDesc none{nullptr, nullptr, nullptr, nullptr};
#ifdef AMP_DEV
    if (phases & 0x100u) k_tile<true, false, false><<<(unsigned)tg.grid, T_WAVES * 64, 0, stream>>>(P, rd, read_base, out, counts, eb, dlist, dcnt, (int)tg.tpb, none, nullptr, nullptr, (uint32_t)tg.grid, ListSrc{} AMP_PHASES_ARG(phases));
    else
#endif
    k_tile<false, false, false><<<(unsigned)tg.grid, T_WAVES * 64, 0, stream>>>(P, rd, read_base, out, counts, eb, dlist, dcnt, (int)tg.tpb, none, nullptr, nullptr, (uint32_t)tg.grid, ListSrc{} AMP_PHASES_ARG(phases));
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Variant 3: the same work as three kernels, cut where the lane mapping changes, so that the
// two phases that need little state (primer clips, window scan) run at a higher occupancy than
// the fused kernel's 4 waves per SIMD:  k_trim (lane = read) -> k_scan (lane = chunk) ->
// k_tile<SPLIT> (quality clip, outputs, counting).
// ---------------------------------------------------------------------------------------
constexpr int S_WAVES = 4;     // waves per block of k_trim / k_scan

__global__ void __launch_bounds__(S_WAVES * 64)
k_trim(KParams P, amp_dev_reads rd, DevOut out, SplitDesc sd) {
    __shared__ uint16_t s_a[S_WAVES][T_MAXOPS * TILE], s_b[S_WAVES][T_MAXOPS * TILE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_u16 *const cigA = (lds_u16 *)s_a[wave], *const cigB = (lds_u16 *)s_b[wave];
    const int64_t i = (int64_t)blockIdx.x * (S_WAVES * 64) + threadIdx.x;
    if (i >= rd.n_reads) return;
    const int32_t pos = rd.pos[i], tlen = rd.tlen[i], lseq = (int32_t)rd.lseq[i];
    const uint32_t flag = rd.flag[i], c0 = rd.cig_off32[i], off8 = rd.seq_off8[i];
    const int ncig = (int)(rd.cig_off32[i + 1] - c0);
    const size_t slot = (size_t)c0 + 3 * (size_t)i;
    const uint8_t *qual = rd.qual + (int64_t)off8 * 8;
    bool defer_full = ncig + 3 > T_MAXOPS || (uint32_t)lseq >= 65536u;
    LdsCig cur{cigA + lane}, tmp{cigB + lane};
    if (!defer_full) {
        uint32_t sumlen = 0;
        for (int k = 0; k < ncig; ++k) { const uint32_t w = rd.cig[c0 + k]; cur.set(k, w); sumlen += w >> 4; }
        if (sumlen >= 4096u) defer_full = true;
    }
    uint32_t meta = defer_full ? SD_DEFER : 0u, lohi = 0;
    TrimState ts{pos, ncig, 0u, 0};
    if (!defer_full) {
        const bool have_qual = lseq > 0 && qual[0] != 0xFF;
        int32_t qs = 0, lo = 0, qlen = 0;
        bool can_q = false;
        if (P.do_trim) {
            trim_primers(P, ts, flag, tlen, lseq, cur, tmp);
            if (!ts.err) can_q = quality_window(ts, lseq, have_qual, cur, qs, lo, qlen);
        }
        if (!ts.err)
            for (int k = 0; k < ts.n; ++k) out.new_cig[slot + k] = cur.get(k);
        lohi = (uint32_t)lo | ((uint32_t)(lo + qlen) << 16);
        meta = (uint32_t)(ts.n & 0xFF) | ((ts.flags & 0xFFu) << 8) | ((uint32_t)(ts.err & 0xFF) << 16) |
               (have_qual ? SD_HAVE_QUAL : 0u) | (can_q ? SD_CAN_Q : 0u) | ((flag & 0x10u) ? SD_REV : 0u);
    }
    sd.pos[i] = ts.pos; sd.lohi[i] = lohi; sd.meta[i] = meta;
}

__global__ void __launch_bounds__(S_WAVES * 64)
k_scan(KParams P, amp_dev_reads rd, SplitDesc sd, uint32_t phases) {
    __shared__ uint32_t s_st[S_WAVES][S_WORDS * TILE];
    __shared__ uint32_t s_map[S_WAVES][T_MAPCAP / 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_u32 *const st = (lds_u32 *)s_st[wave];
    lds_u8 *const cmap = (lds_u8 *)s_map[wave];
    const int64_t i = (int64_t)blockIdx.x * (S_WAVES * 64) + threadIdx.x;
    const bool valid = i < rd.n_reads;
    const int32_t Wd = P.window, mq = P.min_quality;
    const uint32_t mqc = (uint32_t)(mq > 256 ? 256 : mq);
    uint32_t meta = 0, lohi = 0, off8 = 0;
    if (valid) { meta = sd.meta[i]; lohi = sd.lohi[i]; off8 = rd.seq_off8[i]; }
    const int32_t lo = (int32_t)(lohi & 0xFFFFu), hi = (int32_t)(lohi >> 16);
    const bool par_scan = valid && !(meta & SD_DEFER) && (meta & SD_CAN_Q) && Wd <= 8 && (phases & 2u);
    uint32_t nch2 = 0;
    if (par_scan && hi - lo >= Wd) nch2 = (uint32_t)(((hi - Wd) >> 3) - (lo >> 3) + 1);
    uint32_t total2;
    const uint32_t cb2 = wave_excl_scan(nch2, lane, total2);
    st[S_OFF8 * TILE + lane] = off8;
    st[S_LOHI * TILE + lane] = lohi;
    st[S_FF * TILE + lane] = 0xFFFFu;
    st[S_REV * TILE + lane] = (meta & SD_REV) ? 1u : 0u;
    st[S_CB2 * TILE + lane] = cb2;
    const ChunkEnv env{cmap, st, nullptr, nullptr, nullptr, rd.qual, rd.seq, nullptr, 0, 0u, (uint32_t)P.ref_len, mq};
    for (uint32_t base = 0; base < total2; base += T_MAPCAP) {
        wave_sync();
        {
            uint32_t a = cb2 > base ? cb2 : base, b = cb2 + nch2 < base + T_MAPCAP ? cb2 + nch2 : base + T_MAPCAP;
            for (uint32_t c = a; c < b; ++c) cmap[c - base] = (uint8_t)lane;
        }
        wave_sync();
        const uint32_t lim = total2 - base < (uint32_t)T_MAPCAP ? total2 - base : (uint32_t)T_MAPCAP;
        const uint32_t thr = mqc * (uint32_t)Wd;
        switch (Wd) {
            case 1: p2_round<1>(env, lane, lim, base, thr); break; case 2: p2_round<2>(env, lane, lim, base, thr); break;
            case 3: p2_round<3>(env, lane, lim, base, thr); break; case 4: p2_round<4>(env, lane, lim, base, thr); break;
            case 5: p2_round<5>(env, lane, lim, base, thr); break; case 6: p2_round<6>(env, lane, lim, base, thr); break;
            case 7: p2_round<7>(env, lane, lim, base, thr); break; default: p2_round<8>(env, lane, lim, base, thr); break;
        }
    }
    wave_sync();
    if (valid) sd.ff[i] = st[S_FF * TILE + lane];
}

static inline int split_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out,
                               uint32_t *counts, const EventBuf &eb, uint32_t *dlist, uint32_t *dcnt, int n_cu,
                               uint32_t phases, const SplitDesc &sd, hipStream_t stream) {
    if (rd.n_reads == 0) return 0;
    const TileGrid tg = tile_grid(rd.n_reads, n_cu);
    const unsigned g1 = (unsigned)((rd.n_reads + S_WAVES * 64 - 1) / (S_WAVES * 64));
    k_trim<<<g1, S_WAVES * 64, 0, stream>>>(P, rd, out, sd);
    if (P.do_trim && P.window <= 8 && (phases & 2u)) k_scan<<<g1, S_WAVES * 64, 0, stream>>>(P, rd, sd, phases);
    k_tile<false, true, false><<<(unsigned)tg.grid, T_WAVES * 64, 0, stream>>>(P, rd, read_base, out, counts, eb, dlist, dcnt, (int)tg.tpb, sd, nullptr, nullptr, (uint32_t)tg.grid, ListSrc{} AMP_PHASES_ARG(phases));
    return (int)hipGetLastError();
}

}  // namespace amp
