// amp_fast.hpp -- the fast kernel (variant 4, the default): trim + pileup of SIMPLE reads, one lane per
// read, every byte of the read loaded once and kept in registers.  Written for CDNA4 / gfx950.
//
// A read is simple when its CIGAR is one match op covering the whole query ("150M", "150=", ...): nine
// reads in ten of an amplicon run.  For such a read every stage of trim_read (A:426-687) has a closed form
// (SimpleCig in amp_read.hpp: the result is always [S a][M m][S c]) and update_base_counts (A:690-753)
// reduces to "count base q at reference position pos' + (q - a) when qual[q] >= min_quality, a <= q < a+m".
// Everything else -- any other CIGAR, QUAL '*', reads of more than F_NP * 16 bases -- goes on a list that
// the general tile kernel (amp_tile.hpp, k_tile<LIST>) processes afterwards with the exact generic code.
//
// Why one lane per read: the tile kernel maps lanes to 8-base chunks and pays for it per chunk (owner
// look-up, per-read state through LDS, an LDS atomicMin per failing window, a second index map for
// counting): ~280 instructions per chunk against ~20 of actual window arithmetic.  Here all per-read state
// lives in the lane's registers and the only shared structure is the block's window of counters.
//
//   * a wave takes 64 consecutive reads of the coordinate-sorted batch; lane l loads the header of read l,
//     its first CIGAR word and the two primer-table entries, and -- without waiting for those -- its
//     qualities and bases as F_NP 16-base PIECES (16 + 8 bytes each) straight into registers
//   * the pieces are ROTATED per lane: register slot k of lane l holds piece (k + l mod np) mod np of its
//     read.  An amplicon pile has thousands of reads with the same start: in step k lanes then touch
//     different stretches of the reference, and lanes that do share a piece (l, l + np, ...) add into
//     different REPLICAS of the block's LDS window -- per-base LDS atomics without same-address conflicts
//   * sliding-window scan (A:561-649) per piece with v_qsad_pk_u16_u8, first / last failing window as a
//     running min / max in a register; quality clip (A:589-686) in closed form; counting from the same
//     registers: quality flag and count plane by byte-parallel arithmetic, one ds_add_u32 per base with the
//     base's offset as the instruction's immediate
//   * a piece that has a code outside A C G T among its counted bases, or leaves the window, is redone by
//     a careful per-base loop (exact status through the general pass, like the tile kernel does)
#pragma once

#include "amp_tile.hpp"

namespace amp {

constexpr int F_WAVES = 8;            // waves per block (one block per CU: LDS)
constexpr int F_NP = 10;              // 16-base pieces per read held in registers: reads of up to 160 bases
constexpr int F_W = 512;              // reference positions covered by the block's LDS window
constexpr int F_NPL = 4;              // count planes kept in LDS: A C G T (N and '-' never take the fast path)
constexpr int F_STAGE = 10240;        // bytes of a wave's staging buffer = the longest run of quality bytes a tile may span
#ifndef AMP_F_REP
#define AMP_F_REP 4
#endif
#ifndef AMP_F_SKEW
#define AMP_F_SKEW 1
#endif
constexpr int F_REP = AMP_F_REP;              // replicas of the window (lanes that hold the same piece use different ones)
constexpr int F_PLANE = F_W * 4;      // bytes per symbol plane
constexpr int F_REPW = F_NPL * F_W + AMP_F_SKEW;   // words per replica: one word of skew, so that replica r is shifted by r banks

struct FastLds {
    uint4 stage[F_WAVES][F_STAGE / 16 + 2];   // per wave: the tile's quality bytes, then its packed bases (coalesced loads in, rows out)
    uint32_t win[F_REP * F_REPW];
    uint32_t gcount;                  // entries of this block's segment of the general list
};

struct FastGrid { int64_t grid, rpb; };
static inline FastGrid fast_grid(int64_t n_reads, int n_cu) {
    // four blocks per CU (one is resident); a wave gets at least two tiles of 64 reads
    int64_t rpb = (n_reads + 4 * (int64_t)n_cu - 1) / (4 * (int64_t)n_cu);
    rpb = ((rpb + F_WAVES * 64 - 1) / (F_WAVES * 64)) * (F_WAVES * 64);
    if (rpb < 2 * F_WAVES * 64) rpb = 2 * F_WAVES * 64;
    return FastGrid{(n_reads + rpb - 1) / rpb, rpb};
}

// per-byte flag (bit 7) "quality >= mq" of four qualities, for mq <= 128 (mqb = mq in every byte); the host sends
// runs with a larger min_quality to the general kernel
__device__ __forceinline__ uint32_t ok_bits4(uint32_t q, uint32_t mqb) {
    return ((((q & 0x7F7F7F7Fu) | 0x80808080u) - mqb) | q) & 0x80808080u;
}

// 16-bit mask -> byte flags (bit 0 of byte i = bit 4 d + i of m), one dword of a piece
__device__ __forceinline__ uint32_t nibble_to_bytes(uint32_t m, int d) {
    // bit i of the nibble times 2^(7 i) lands on bit 8 i; the cross terms land between the byte positions
    return (((m >> (4 * d)) & 15u) * 0x00204081u) & 0x01010101u;
}

// the 16 base codes of a piece as bytes in base order (seq holds two bases per byte, high nibble first)
__device__ __forceinline__ void spread_codes(const uint2 &s, uint32_t (&cb)[4]) {
    const uint32_t e0 = (s.x >> 4) & 0x0F0F0F0Fu, o0 = s.x & 0x0F0F0F0Fu;     // bases 0 2 4 6 | 1 3 5 7
    const uint32_t e1 = (s.y >> 4) & 0x0F0F0F0Fu, o1 = s.y & 0x0F0F0F0Fu;     // bases 8 10 12 14 | 9 11 13 15
    cb[0] = __builtin_amdgcn_perm(o0, e0, 0x05010400u);
    cb[1] = __builtin_amdgcn_perm(o0, e0, 0x07030602u);
    cb[2] = __builtin_amdgcn_perm(o1, e1, 0x05010400u);
    cb[3] = __builtin_amdgcn_perm(o1, e1, 0x07030602u);
}

// bit 7 of a byte set when its code (0..15) is not one of A C G T (1 2 4 8): not exactly one bit set
__device__ __forceinline__ uint32_t not_acgt(uint32_t cb) {
    const uint32_t m = (cb | 0x80808080u) - 0x01010101u;      // n - 1 per byte; bit 7 survives unless n == 0
    const uint32_t t = cb & m & 0x0F0F0F0Fu;                   // n & (n - 1)
    return ((t + 0x7F7F7F7Fu) | ~m) & 0x80808080u;
}

// count-plane number (0..3) of A C G T codes per byte; some plane 0..3 for any other code
__device__ __forceinline__ uint32_t col_bytes(uint32_t cb) {
    return (((cb >> 1) & 0x07070707u) - ((cb >> 3) & 0x01010101u)) & 0x03030303u;
}

// 16 failing-window bits of one piece: bit b set <=> the W-byte window starting at byte b of q (continued in nx) sums to < thr
template <int W>
__device__ __forceinline__ uint32_t piece_fail_bits(const uint4 &q, const uint2 &nx, uint32_t thr) {
    return window_fail_bits16<W>(make_uint2(q.x, q.y), make_uint2(q.z, q.w), thr) |
           (window_fail_bits16<W>(make_uint2(q.z, q.w), nx, thr) << 8);
}

// in-kernel phase stamps (development builds only; the numbers are shares, not durations)
#ifdef AMP_DEV
#define F_STAMP_DECL unsigned long long f_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f_prev = __builtin_amdgcn_s_memtime()
#define F_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); unsigned long long f_n = __builtin_amdgcn_s_memtime(); f_t[k] += f_n - f_prev; f_prev = f_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define F_STAMP_VM(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0x0070); unsigned long long f_n = __builtin_amdgcn_s_memtime(); f_t[k] += f_n - f_prev; f_prev = f_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define F_STAMP_OUT do { if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&ctr[8 + k], f_t[k]); } while (0)
#else
#define F_STAMP_DECL
#define F_STAMP(k)
#define F_STAMP_VM(k)
#define F_STAMP_OUT
#endif

template <int W>
__global__ void __launch_bounds__(F_WAVES * 64, 2)
k_fast(KParams P, amp_dev_reads rd, DevOut out, uint32_t *counts, unsigned long long *ctr, uint32_t *glist, uint32_t *gcnt,
       int reads_per_block) {
    __shared__ FastLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t n = rd.n_reads;
    const int64_t rb = (int64_t)blockIdx.x * reads_per_block;
    const int64_t re = rb + reads_per_block < n ? rb + reads_per_block : n;
    lds_u32 *const win = (lds_u32 *)L.win;
    for (int i = tid; i < F_REP * F_REPW; i += F_WAVES * 64) win[i] = 0;
    if (tid == 0) L.gcount = 0;
    // the block's window: anchored 16 positions left of its first read (sorted input: nothing of this block starts
    // left of that read; the margin keeps the piece that straddles a read's first counted base inside the window)
    int32_t win_base = rb < n ? rd.pos[rb] : 0;
    win_base = (win_base < 16 ? 0 : win_base - 16) & ~15;
    uint32_t wlim;
    {
        const int64_t lim = (int64_t)P.ref_len - win_base;
        wlim = lim <= 0 ? 0u : (lim > F_W ? (uint32_t)F_W : (uint32_t)lim);
    }
    __syncthreads();

    const int32_t mq = P.min_quality;
    const uint32_t mqc = (uint32_t)(mq > 256 ? 256 : mq);
    const uint32_t thr = mqc * (uint32_t)W;
    const uint32_t mqb = (uint32_t)mq * 0x01010101u;             // mq <= 128 (fast_launch checks)
    const uint32_t G = (uint32_t)P.ref_len;
    const int64_t per_wave = reads_per_block / F_WAVES;
    const int64_t wbeg = rb + (int64_t)wave * per_wave;
    int64_t wend = wbeg + per_wave;
    wend = wend < re ? wend : re;
    unsigned long long n_err = 0;
    // Bank plan of the counting adds.  In an amplicon pile the 64 reads of a wave start at the same position; in
    // step k lane l then adds at word C + 16 p + b with p = (k + l) mod np: for a fixed b that is two banks for all
    // lanes.  So: lanes 8 apart use different REPLICAS (replica r is skewed by r banks; lanes l and l + np, which
    // hold the same piece, are also at least 8 apart), and the four 4-base groups of a piece are rotated by
    // rb = 4 * ((l >> 1) & 3) positions inside the piece (group g of the lane adds at offsets ((4 g + rb) & 15) ..+3):
    // parity of p (2) x rb (4) x replica skew (4) = 32 banks for the 32 lanes serviced together.  The rotation is a
    // per-lane constant folded into four base registers; each add keeps its offset 4 b as an immediate.
    const uint32_t rep = ((uint32_t)lane >> 3) & (uint32_t)(F_REP - 1);
    const uint32_t rotb = 4u * (((uint32_t)lane >> 1) & 3u);
    int32_t gdelta[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gdelta[g] = 0 * (int32_t)rotb;   // TEMP: rotation needs the data rotated too

    lds_u8 *const stage = (lds_u8 *)L.stage[wave];
    F_STAMP_DECL;

    // ---- software pipeline: while tile t is computed from registers, the bytes of tile t + 1 are on their way --
    // (qualities by LDS-DMA straight into the wave's staging buffer, which is idle once the rows of tile t are
    // in registers; packed bases, first CIGAR word and primer-table entries into registers; the header of
    // tile t + 2 into registers)
    struct Hdr { int32_t pos, tlen; uint32_t lseq, flag, c0, c1, o8; };
    struct Geo { uint32_t np, row, Tq; int ntake; bool solo, taken, shortq, fastq, in_ref; };
    struct L2 { uint32_t w0; int32_t tabL, tabR; uint2 raws[F_STAGE / 1024];
#ifdef AMP_F_NODMA
        uint2 rawq[F_STAGE / 512];
#endif
    };
    auto load_hdr = [&](int64_t t0) {
        Hdr h{0, 0, 0u, 0u, 0u, 0u, 0u};
        const int64_t i = t0 + lane;
        if (i < wend) {
            h.pos = rd.pos[i]; h.flag = rd.flag[i]; h.tlen = rd.tlen[i]; h.lseq = rd.lseq[i];
            h.c0 = rd.cig_off32[i]; h.c1 = rd.cig_off32[i + 1]; h.o8 = rd.seq_off8[i];
        }
        return h;
    };
    // the tile: the leading reads whose bytes form one run of at most F_STAGE quality bytes
    auto geometry = [&](const Hdr &h, int64_t t0, uint32_t &m0) {
        Geo g;
        const bool valid = t0 + lane < wend;
        g.shortq = valid && h.lseq >= 1u && h.lseq <= (uint32_t)(F_NP * 16);
        g.np = g.shortq ? (h.lseq + 15u) >> 4 : 1u;
        m0 = __builtin_amdgcn_readfirstlane(h.o8);
        g.row = (h.o8 - m0) * 8u;                                            // byte offset of the read's qualities in the run
        const uint32_t nch = (h.lseq + 7u) >> 3;
        // a row is read as np pieces of 16 bytes: up to 8 bytes past the read's own padded bytes
        const bool fits = valid && h.o8 >= m0 && (h.o8 - m0) <= (uint32_t)(F_STAGE / 8) && g.row + 16u * g.np <= (uint32_t)F_STAGE + 8u &&
                          (g.shortq || g.row + 8u * nch <= (uint32_t)F_STAGE);
        const unsigned long long fitmask = __ballot(fits);
        g.ntake = fitmask == ~0ull ? 64 : __builtin_ctzll(~fitmask);
        g.solo = g.ntake == 0;                                                // the first read alone is too long: general pass
        if (g.solo) g.ntake = 1;
        g.taken = lane < g.ntake && !g.solo;
        g.Tq = g.solo ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)(g.row + 8u * nch), g.ntake - 1);   // bytes of the run (scalar)
        g.Tq = g.Tq > (uint32_t)F_STAGE ? (uint32_t)F_STAGE : g.Tq;
        g.fastq = g.taken && g.shortq;
        g.in_ref = (uint32_t)h.pos < G && (uint32_t)(h.pos + (int32_t)h.lseq - 1) < G;          // A:450-451
        return g;
    };
    // second level of loads: first CIGAR word, the two primer-table entries of A:450-451 (for a simple read they
    // depend on the header only), the tile's quality bytes by LDS-DMA (lane l moves bytes [1024 s + 16 l, + 16) of
    // the run to the same offset of the staging buffer) and its packed bases (8 bytes per lane and load)
    auto issue_l2 = [&](const Hdr &h, const Geo &g, uint32_t m0) {
        L2 x;
        x.w0 = 0; x.tabL = -1; x.tabR = -1;
        if (g.fastq && h.c1 > h.c0) x.w0 = rd.cig[h.c0];
        if (g.fastq && P.do_trim && g.in_ref) { x.tabL = P.max_end[h.pos]; x.tabR = P.min_start[h.pos + (int32_t)h.lseq - 1]; }
        const uint8_t *qrun = rd.qual + (int64_t)m0 * 8;
        const uint8_t *srun = rd.seq + (int64_t)m0 * 4;
        // lanes past the run re-read its end
        const uint32_t lastq = g.Tq ? (g.Tq - 1u) & ~15u : 0u, lasts = g.Tq ? ((g.Tq >> 1) - 1u) & ~7u : 0u;
#ifdef AMP_F_NODMA
        const uint32_t lastq8 = g.Tq ? (g.Tq - 1u) & ~7u : 0u;
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 512; ++sl) {
            uint32_t off = (uint32_t)(sl * 512 + lane * 8);
            off = off < lastq8 ? off : lastq8;
            x.rawq[sl] = *(const uint2 *)(qrun + off);
        }
#else
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 1024; ++sl) {
            uint32_t off = (uint32_t)(sl * 1024 + lane * 16);
            off = off < lastq ? off : lastq;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qrun + off),
                                             (__attribute__((address_space(3))) void *)(stage + sl * 1024), 16, 0, 0);
        }
#endif
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 1024; ++sl) {
            uint32_t off = (uint32_t)(sl * 512 + lane * 8);
            off = off < lasts ? off : lasts;
            const uint32_t *sp = (const uint32_t *)(srun + off);
            x.raws[sl] = make_uint2(sp[0], sp[1]);
        }
        return x;
    };

    // Results of a tile are STORED ONE TILE LATER, right behind the wait at the top of the loop: stores and loads
    // retire through one in-order counter, so a store issued at the end of a tile would make that wait
    // last until the store has reached memory (measured: 30 % of the kernel).
    struct Pend { int64_t i; uint32_t slot_lo; int32_t pos, reflen; uint32_t ncig, cw0, cw1, cw2, status, flags, entry; bool simple, has; };
    Pend pend{0, 0u, 0, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, false, false};
    auto store_pending = [&](const Pend &r) {
        if (r.simple) {
            uint32_t *home = out.new_cig + ((size_t)r.slot_lo + 3 * (size_t)r.i);
            if (r.ncig > 0u) home[0] = r.cw0;
            if (r.ncig > 1u) home[1] = r.cw1;
            if (r.ncig > 2u) home[2] = r.cw2;
            if (out.new_pos) out.new_pos[r.i] = r.pos;
            if (out.new_ncig) out.new_ncig[r.i] = r.ncig;
            if (out.ref_len) out.ref_len[r.i] = r.reflen;
            if (out.trim_flags) out.trim_flags[r.i] = (uint8_t)r.flags;
            if (out.status) out.status[r.i] = (uint8_t)r.status;
        }
        // hand-over to the general pass: one reservation per wave
        const unsigned long long m = __ballot(r.has);
        if (m) {
            uint32_t gb = 0;
            if (lane == 0) gb = __hip_atomic_fetch_add((lds_u32 *)&L.gcount, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            gb = __shfl(gb, 0);
            if (r.has) glist[(size_t)rb + gb + __popcll(m & ((1ull << lane) - 1ull))] = r.entry;
        }
    };

    int64_t i0 = wbeg, i1 = wbeg;
    Hdr hA = load_hdr(i0), hB{0, 0, 0u, 0u, 0u, 0u, 0u};
    uint32_t m0A = 0, m0B = 0;
    Geo gA = geometry(hA, i0, m0A), gB = gA;
    L2 xA{}, xB{};
    xA = issue_l2(hA, gA, m0A);
    i1 = i0 + gA.ntake;
    hB = load_hdr(i1);
    while (i0 < wend) {
        const int64_t i = i0 + lane;
        const Hdr h = hA;
        const Geo g = gA;
        const int32_t pos = h.pos, tlen = h.tlen;
        const uint32_t lseq = h.lseq, flag = h.flag, c0 = h.c0, c1 = h.c1, o8 = h.o8;
        const uint32_t np = g.np;
        const bool solo = g.solo, taken = g.taken, fastq = g.fastq, in_ref = g.in_ref;
        const int ntake = g.ntake;
        const uint32_t w0 = xA.w0;
        const int32_t tabL = xA.tabL, tabR = xA.tabR;
        // ---- rows: slot k of the lane holds piece (k + rot) mod np of its read (slots >= np: a copy of the last
        // piece and an index past the read, which every range test below excludes) -------------------------------
        const uint32_t rot = (uint32_t)lane % np;
        const uint8_t *qrow = rd.qual + (int64_t)o8 * 8;
        const uint8_t *srow = rd.seq + (int64_t)o8 * 4;
        const uint32_t lrow = fastq ? g.row : 0u;
        uint4 q16[F_NP];
        uint2 s8[F_NP];
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the DMA of this tile's qualities has landed, its other loads too
        F_STAMP(1);
#ifdef AMP_F_NODMA
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 512; ++sl) *(lds_u32x2 *)(stage + sl * 512 + lane * 8) = amp_u32x2{xA.rawq[sl].x, xA.rawq[sl].y};
#endif
        wave_sync();
#pragma unroll
        for (int k = 0; k < F_NP; ++k) {
            uint32_t p = (uint32_t)k + rot;
            p = p >= np ? p - np : p;
            p = (uint32_t)k < np ? p : np - 1u;
            const lds_u8 *src = stage + lrow + p * 16u;
            const amp_u32x2 a = *(const lds_u32x2 *)src, b = *(const lds_u32x2 *)(src + 8);
            q16[k] = make_uint4(a.x, a.y, b.x, b.y);
        }
        wave_sync();
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 1024; ++sl) *(lds_u32x2 *)(stage + sl * 512 + lane * 8) = amp_u32x2{xA.raws[sl].x, xA.raws[sl].y};
        wave_sync();
#pragma unroll
        for (int k = 0; k < F_NP; ++k) {
            uint32_t p = (uint32_t)k + rot;
            p = p >= np ? p - np : p;
            p = (uint32_t)k < np ? p : np - 1u;
            const lds_u8 *src = stage + (lrow >> 1) + p * 8u;
            s8[k] = make_uint2(*(const lds_u32 *)src, *(const lds_u32 *)(src + 4));
        }
        wave_sync();                                 // every lane has its rows: the staging buffer may be overwritten
        store_pending(pend);                         // the previous tile's results
        // ---- primer clips in closed form (A:450-558) ---------------------------------------------------------------
        const bool simple = fastq && c1 - c0 == 1u && is_simple_cigar(1, w0, (int32_t)lseq);
        const bool rev = (flag & 0x10u) != 0;
        TrimState ts{pos, 1, 0u, 0};
        SimpleCig sc{w0 & 15u, 0, (int32_t)lseq, 0};
        if (simple && P.do_trim) {
            if (!in_ref) ts.err = AMP_RS_INDEX_REF;
            else trim_primers_simple_tab(P, ts, flag, tlen, (int32_t)lseq, sc, tabL, tabR);
        }
        const bool scan = simple && P.do_trim && !ts.err;
        const int32_t lo = !scan ? 0 : (sc.m > 0 ? sc.a : (int32_t)lseq), qlen = scan ? sc.m : 0, hi = lo + qlen;
        // the 3' end's shrinking windows (A:575-576, A:637-638) need at most W-1 bytes; loaded by every lane (from
        // the start of its read when there is nothing to scan): a load under a branch is waited for at its end
        const int32_t first = !scan ? 0 : (rev || qlen < W) ? lo : lo + qlen - W + 1;
        const int32_t tab = first & ~7;
        const uint2 tw0 = *(const uint2 *)(qrow + tab), tw1 = *(const uint2 *)(qrow + tab + 8);
        // ---- next tile: its bytes start moving now, its header was loaded during the previous tile.  No branch
        // around these loads (behind the wave's last tile they fetch the first bytes of the batch): a branch would
        // make the compiler wait for everything in flight at its end ------------------------------------------------
        gB = geometry(hB, i1, m0B);
        xB = issue_l2(hB, gB, m0B);
        const int64_t i2 = i1 + gB.ntake;
        const Hdr hC = load_hdr(i2);
        F_STAMP(2);          // staged, rows in registers, primer clips, next tile issued

        // ---- sliding-window scan: first failing window start (forward) / last failing window end (reverse) --
        int32_t ffmin = 0x7FFFFFFF, lemax = -1;
        if (P.do_trim) {
#pragma unroll
            for (int k = 0; k < F_NP; ++k) {
                uint32_t p = (uint32_t)k + rot;
                p = p >= np ? p - np : p;
                p = (uint32_t)k < np ? p : np;
                const int32_t j0 = (int32_t)(p * 16u);
                // the next piece's first bytes: slot k + 1, or slot 0 behind the lane's last slot (behind the read's
                // last piece that is the wrong piece, but no window that is looked at reaches it)
                uint2 nx = make_uint2(q16[0].x, q16[0].y);
                if (k + 1 < F_NP && (uint32_t)(k + 1) < np) nx = make_uint2(q16[k + 1].x, q16[k + 1].y);
                uint32_t fail = piece_fail_bits<W>(q16[k], nx, thr);
                int32_t blo = lo - j0, bhi = hi - W - j0;                   // window starts j0+b must lie in [lo, hi - W]
                blo = blo < 0 ? 0 : (blo > 16 ? 16 : blo); bhi = bhi > 15 ? 15 : (bhi < -1 ? -1 : bhi);
                fail &= (0xFFFFu >> (15 - bhi)) & (0xFFFFu << blo);
                const int32_t f1 = j0 + (__builtin_ffs((int)fail) - 1), e1 = j0 + (31 - __builtin_clz(fail)) + W;
                ffmin = fail && f1 < ffmin ? f1 : ffmin;
                lemax = fail && e1 > lemax ? e1 : lemax;
            }
        }

        F_STAMP(3);          // primer clips + window scan
        // ---- quality clip, outputs (A:589-686) ------------------------------------------------------------------
        bool general = (taken || (solo && lane == 0)) && !simple;
        bool counted = false, stored = false;
        uint32_t ncig = 0, cw[3] = {0u, 0u, 0u};
        int32_t reflen = 0;
        if (simple) {
            // the read's first quality byte (0xFF = QUAL '*') sits in the slot that holds piece 0: slot (np - rot) mod np
            uint32_t fb = q16[0].x;
#pragma unroll
            for (int k = 1; k < F_NP; ++k) fb = ((uint32_t)k + rot == np) ? q16[k].x : fb;
            if ((fb & 0xFFu) == 0xFFu) {
                general = true;                   // QUAL '*': the generic code reports it (A:561-562, A:718)
            } else {
                if (scan) {
                    int32_t iq;
                    if (!rev && ffmin != 0x7FFFFFFF) iq = ffmin - lo;
                    else if (rev && lemax >= 0) iq = lemax - lo;
                    else {
                        // no full window failed: the shrinking windows at the 3' end decide
                        iq = rev ? 0 : qlen;
                        int32_t acc = 0;
                        const uint64_t t_lo = (uint64_t)tw0.x | ((uint64_t)tw0.y << 32), t_hi = (uint64_t)tw1.x | ((uint64_t)tw1.y << 32);
                        const int32_t kmax = qlen < W - 1 ? qlen : W - 1;
                        for (int32_t k = 1; k <= kmax; ++k) {
                            const uint32_t o = (uint32_t)((rev ? lo + k - 1 : hi - k) - tab);      // 0..15
                            acc += (int32_t)(((o < 8u ? t_lo : t_hi) >> ((o & 7u) * 8u)) & 0xFFu);
                            if ((int64_t)acc < (int64_t)mq * k) iq = rev ? k : qlen - k;
                        }
                    }
                    trim_quality_apply_simple(ts, rev, iq, qlen, sc);
                }
                if (!ts.err) {
                    if (sc.a > 0) cw[ncig++] = ((uint32_t)sc.a << 4) | OP_S;
                    if (sc.m > 0) cw[ncig++] = ((uint32_t)sc.m << 4) | sc.op;
                    if (sc.c > 0) cw[ncig++] = ((uint32_t)sc.c << 4) | OP_S;
                    reflen = sc.m > 0 ? sc.m : 1;
                }
                stored = true;
                if (ts.err) ++n_err;
                counted = !ts.err && P.do_count;
            }
        }

        F_STAMP(4);          // quality clip, outputs
        // ---- counting (A:709-753 for a read without indels) -----------------------------------------------------
        uint32_t redo = 0;                        // pieces (slots) the careful loop has to do
        const int32_t qa = counted ? sc.a : 0, qb = counted ? sc.a + sc.m : 0;
        if (P.do_count) {
            const int32_t dbase = ts.pos - win_base - qa;              // window offset of query base 0
            lds_u8 *const wrep = (lds_u8 *)win + rep * (uint32_t)(F_REPW * 4);
#pragma unroll
            for (int k = 0; k < F_NP; ++k) {
                uint32_t p = (uint32_t)k + rot;
                p = p >= np ? p - np : p;
                p = (uint32_t)k < np ? p : np;
                const int32_t j0 = (int32_t)(p * 16u);
                int32_t klo = qa - j0, khi = qb - j0;
                klo = klo < 0 ? 0 : klo; khi = khi > 16 ? 16 : khi;
                if (khi > klo) {                                            // some base of the piece is counted
                    const int32_t d0 = dbase + j0;                          // window offset of the piece's base 0
                    const uint32_t rng = ((1u << khi) - 1u) & ~((1u << klo) - 1u);   // khi <= 16
                    // per base (byte): bit 0 = counted (quality and range), codes, validity, plane
                    uint32_t f[4], cb[4], col[4];
                    f[0] = (ok_bits4(q16[k].x, mqb) >> 7) & nibble_to_bytes(rng, 0);
                    f[1] = (ok_bits4(q16[k].y, mqb) >> 7) & nibble_to_bytes(rng, 1);
                    f[2] = (ok_bits4(q16[k].z, mqb) >> 7) & nibble_to_bytes(rng, 2);
                    f[3] = (ok_bits4(q16[k].w, mqb) >> 7) & nibble_to_bytes(rng, 3);
                    spread_codes(s8[k], cb);
                    uint32_t bad = 0;
#pragma unroll
                    for (int d = 0; d < 4; ++d) { bad |= (not_acgt(cb[d]) >> 7) & f[d]; col[d] = col_bytes(cb[d]); }
                    // (a counted N is rare -- N calls come with low qualities -- and takes the careful loop too)
                    const bool safe = bad == 0u && wlim >= 16u && (uint32_t)d0 <= wlim - 16u;
                    if (safe) {
                        lds_u8 *const wb = wrep + (uint32_t)d0 * 4u;
                        lds_u8 *const wbg[4] = {wb + gdelta[0], wb + gdelta[1], wb + gdelta[2], wb + gdelta[3]};
#pragma unroll
                        for (int b = 0; b < 16; ++b) {
                            const uint32_t plane = ((col[b >> 2] >> (8 * (b & 3))) & 3u) << 11;      // F_PLANE = 2048
                            lds_add((lds_u32 *)(wbg[b >> 2] + plane + 4 * b), (f[b >> 2] >> (8 * (b & 3))) & 1u);
                        }
                    } else {
                        redo |= 1u << k;
                    }
                }
            }
        }
        F_STAMP(5);          // counting
        if (__ballot(redo != 0u)) {
            // careful loop (rare): bases of the flagged pieces one by one, straight from memory
            bool bad = false;
            if (redo) {
                for (int k = 0; k < F_NP; ++k) {
                    if (!((redo >> k) & 1u)) continue;
                    uint32_t p = (uint32_t)k + rot;
                    p = p >= np ? p - np : p;
                    const int32_t j0 = (int32_t)(p * 16u);
                    for (int32_t q = j0 < qa ? qa : j0; q < j0 + 16 && q < qb; ++q) {
                        if ((int32_t)qrow[q] < mq) continue;
                        const uint32_t sb = srow[q >> 1];
                        const uint32_t col = col_of_code((q & 1) ? (sb & 15u) : (sb >> 4));
                        const int32_t rp = ts.pos + (q - qa);
                        const uint32_t d = (uint32_t)(rp - win_base);
                        if (col > 4u || (uint32_t)rp >= G) bad = true;
                        else if (d < (uint32_t)F_W && col < (uint32_t)F_NPL) lds_add(win + rep * (uint32_t)F_REPW + col * F_W + d, 1u);
                        else atomicAdd(&counts[(size_t)rp * AMP_NSYM + col], 1u);
                    }
                }
            }
            redo = bad ? 1u : 0u;
        } else {
            redo = 0u;
        }

        // ---- results and hand-over to the general pass: kept for the next turn of the loop ---------------------------
        {
            uint32_t entry = 0;
            bool has = false;
            if (general) { entry = (uint32_t)i; has = true; }
            else if (counted && redo) { entry = (uint32_t)i | GL_STATUS_ONLY; has = true; }   // a base could not be counted: exact status wanted
            pend = Pend{i, c0, ts.pos, reflen, ncig, cw[0], cw[1], cw[2], (uint32_t)ts.err, ts.err ? 0u : ts.flags, entry, stored, has};
        }
        F_STAMP(6);          // careful loop, hand-over
        hA = hB; gA = gB; xA = xB; m0A = m0B; hB = hC;
        i0 = i1; i1 = i2;
    }
    store_pending(pend);

    __syncthreads();
    for (int i = tid; i < F_NPL * F_W; i += F_WAVES * 64) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < F_REP; ++r) v += win[r * F_REPW + i];
        if (v) {
            const int sym = i / F_W, d = i - sym * F_W;
            atomicAdd(&counts[(size_t)(win_base + d) * AMP_NSYM + sym], v);
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
    F_STAMP_OUT;
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

static inline int fast_launch(const KParams &P, const amp_dev_reads &rd, const DevOut &out, uint32_t *counts, unsigned long long *ctr,
                              uint32_t *glist, uint32_t *gcnt, const FastGrid &fg, hipStream_t stream) {
    const unsigned g = (unsigned)fg.grid, t = F_WAVES * 64;
    const int rpb = (int)fg.rpb;
    switch (P.window) {
        case 1: k_fast<1><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        case 2: k_fast<2><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        case 3: k_fast<3><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        case 4: k_fast<4><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        case 5: k_fast<5><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        case 6: k_fast<6><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        case 7: k_fast<7><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
        default: k_fast<8><<<g, t, 0, stream>>>(P, rd, out, counts, ctr, glist, gcnt, rpb); break;
    }
    return (int)hipGetLastError();
}

}  // namespace amp
