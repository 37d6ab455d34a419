// amp_fast.hpp -- the fast kernel (variant 4, the default): trim + pileup of SIMPLE reads, one lane per
// read, every byte of the batch loaded once.  Written for CDNA4 / gfx950.
//
// A read is simple when its CIGAR is one match op covering the whole query ("150M", "150=", ...): nine
// reads in ten of an amplicon run.  For such a read every stage of trim_read (A:426-687) has a closed form
// (SimpleCig in amp_read.hpp: the result is always [S a][M m][S c]) and update_base_counts (A:690-753)
// reduces to "count base q at reference position pos' + (q - a) when qual[q] >= min_quality, a <= q < a+m".
// Everything else -- any other CIGAR, QUAL '*', reads of more than ~150 bases -- goes on a list that
// the general tile kernel (amp_tile.hpp, k_tile<LIST>) processes afterwards with the exact generic code.
//
// What bounds this work on MI355X is the vector ALU: integer VALU instructions issue at one per FOUR cycles
// per SIMD (tools/micro/valu_rate.hip), so the design minimises instructions per base:
//   * one lane per read: all per-read state lives in registers, no owner look-ups or per-chunk hand-offs
//     through LDS (the tile kernel spends ~280 instructions per 8-base chunk on those)
//   * a wave takes 64 consecutive reads of the sorted batch (a TILE).  Their quality bytes form one run of
//     memory: LDS-DMA (global_load_lds_dwordx4, coalesced, no registers) drops it into the wave's staging
//     buffer and every lane reads its own read back as 16-base PIECES; the packed bases follow through the
//     same buffer.  The whole thing is software-pipelined: tile t + 1 is in flight while tile t is computed
//     from registers, and tile t's results are stored one turn later (stores and loads retire through ONE
//     in-order counter: a late store would stall the next wait for loads)
//   * sliding-window scan (A:561-649) per piece with v_qsad_pk_u16_u8, first / last failing window as a
//     running min / max in a register; primer and quality clips (A:450-558, A:589-686) in closed form
//   * counting: every wave owns a small window of PACKED counters in LDS -- one 32-bit word per reference
//     position, one byte per base A C G T -- so that a base costs ONE vector instruction (an SDWA shift that
//     turns its "counted" byte into 1 << 8 col) and one ds_add_u32 whose address is the piece's base
//     register plus an immediate.  The bytes cannot overflow: a counter gets at most 16 increments per
//     tile and the window is folded into the block's 32-bit window every 15 tiles at the latest.
//   * bank plan: an amplicon pile has thousands of reads with the same start.  Lane l works on piece
//     (k + l) mod np in step k, starts its pieces 8 bases early when bit 1 of l is set, and adds into
//     replica (l >> 2) & 7 of the window (replica r is skewed by r banks): the 32 lanes serviced together
//     hit 32 different banks, lanes that hold the same piece hit different replicas
//   * a piece that has a code outside A C G T among its counted bases, or leaves the window, is redone by
//     a careful per-base loop (exact status through the general pass, like the tile kernel does)
#pragma once

#include "amp_tile.hpp"

namespace amp {

constexpr int F_WAVES = 8;            // waves per block (one block per CU: LDS)
constexpr int F_NP = 10;              // 16-base pieces per read held in registers
constexpr int F_MAXLEN = 152;         // longest read the fast path takes: F_NP pieces must cover it from 8 bases before its start
constexpr int F_PW = 256;             // reference positions covered by a wave's packed window
constexpr int F_REP = 8;              // replicas of the packed window
constexpr int F_REPW = F_PW + 1;      // words per replica: one word of skew, so that replica r is shifted by r banks
constexpr int F_BW = 512;             // reference positions covered by the block's 32-bit window
constexpr int F_NPL = 4;              // its planes: A C G T (N and '-' never take the fast path)
constexpr int F_STAGE = 10240;        // bytes of a wave's staging buffer = the longest run of quality bytes a tile may span
constexpr int F_PAD = 16;             // bytes in front of the staged run (rows that start 8 bases early)
constexpr int F_FLUSH = 15;           // tiles between two folds of a packed window (16 increments per counter and tile at most)

struct FastLds {
    uint4 stage[F_WAVES][(F_PAD + F_STAGE + 16) / 16];   // per wave: the tile's quality bytes, then its packed bases
    uint32_t pwin[F_WAVES][F_REP * F_REPW];              // per wave: packed counters, byte c of a word = base c (A C G T)
    uint32_t bwin[F_NPL * F_BW];                         // the block's window, 32-bit counters
};

struct FastGrid { int64_t grid, rpb; };
static inline FastGrid fast_grid(int64_t n_reads, int n_cu) {
#ifndef AMP_F_BPC
#define AMP_F_BPC 1
#endif
    // AMP_F_BPC blocks per CU (one is resident); a wave gets at least two tiles of 64 reads
    int64_t rpb = (n_reads + AMP_F_BPC * (int64_t)n_cu - 1) / (AMP_F_BPC * (int64_t)n_cu);
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

// per byte: 8 * plane number of an A C G T code in bits 3..4 (bits 5..7 may hold anything: a shift count uses 5 bits)
__device__ __forceinline__ uint32_t shift_bytes(uint32_t cb) {
    const uint32_t t = cb & 0x08080808u;                         // T: both bits
    return ((cb << 2) & 0x18181818u) | t | (t << 1);             // C -> 8, G -> 16, T -> 24
}

// (byte J of val) << (byte J of sh): one SDWA instruction
template <int J>
__device__ __forceinline__ uint32_t shl_byte(uint32_t sh, uint32_t val) {
    uint32_t r;
    if (J == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(val));
    if (J == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(val));
    if (J == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(val));
    if (J == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(val));
    return r;
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
    lds_u32 *const bwin = (lds_u32 *)L.bwin;
    lds_u32 *const pwin = (lds_u32 *)L.pwin[wave];
    for (int i = tid; i < F_NPL * F_BW; i += F_WAVES * 64) bwin[i] = 0;
    for (int i = lane; i < F_REP * F_REPW; i += 64) pwin[i] = 0;
    // the block's window: anchored 16 positions left of its first read (sorted input: nothing of this block starts
    // left of that read)
    int32_t bw_base = rb < n ? rd.pos[rb] : 0;
    bw_base = (bw_base < 16 ? 0 : bw_base - 16) & ~15;
    __syncthreads();

    const int32_t mq = P.min_quality;
    const uint32_t mqc = (uint32_t)(mq > 256 ? 256 : mq);
    const uint32_t thr = mqc * (uint32_t)W;
    const uint32_t mqb = (uint32_t)mq * 0x01010101u;             // mq <= 128 (the host sends other runs to the general kernel)
    const uint32_t G = (uint32_t)P.ref_len;
    const int64_t per_wave = reads_per_block / F_WAVES;
    const int64_t wbeg = rb + (int64_t)wave * per_wave;
    int64_t wend = wbeg + per_wave;
    wend = wend < re ? wend : re;
    unsigned long long n_err = 0;
    // lane constants of the bank plan (see the head of this file)
    const uint32_t rep = ((uint32_t)lane >> 2) & (uint32_t)(F_REP - 1);
    const uint32_t phi_lane = ((uint32_t)lane >> 1) & 1u ? 8u : 0u;
    lds_u8 *const wrep = (lds_u8 *)pwin + rep * (uint32_t)(F_REPW * 4);
    lds_u8 *const stage = (lds_u8 *)L.stage[wave] + F_PAD;          // the run starts here
    int32_t pw_base = 0;                                            // anchor of the wave's packed window
    int pw_tiles = F_FLUSH;                                         // tiles added since the last fold (forces an anchor for the first tile)
    F_STAMP_DECL;

    // folds the wave's packed window into the block's 32-bit window (or the global table) and clears it
    auto fold = [&]() {
        wave_sync();
#pragma unroll 1
        for (int idx = lane; idx < F_PW; idx += 64) {
            uint32_t ag = 0, ct = 0;                                 // A | G << 16, C | T << 16
#pragma unroll
            for (int r = 0; r < F_REP; ++r) {
                const uint32_t w = pwin[r * F_REPW + idx];
                pwin[r * F_REPW + idx] = 0;
                ag += w & 0x00FF00FFu; ct += (w >> 8) & 0x00FF00FFu;
            }
            if (ag | ct) {
                const int32_t p = pw_base + idx;
                const uint32_t d = (uint32_t)(p - bw_base);
                const uint32_t c4[4] = {ag & 0xFFFFu, ct & 0xFFFFu, ag >> 16, ct >> 16};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (!c4[c]) continue;
                    if (d < (uint32_t)F_BW) lds_add(bwin + c * F_BW + d, c4[c]);
                    else if ((uint32_t)p < G) atomicAdd(&counts[(size_t)p * AMP_NSYM + c], c4[c]);
                }
            }
        }
        wave_sync();
    };

    // ---- software pipeline: while tile t is computed from registers, the bytes of tile t + 1 are on their way --
    struct Hdr { int32_t pos, tlen; uint32_t lseq, flag, c0, c1, o8; };
    struct Geo { uint32_t np, phi, row, Tq; int ntake; bool solo, taken, fastq, in_ref; };
    struct L2 { uint32_t w0; int32_t tabL, tabR; uint2 raws[F_STAGE / 1024]; };
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
        const bool shortq = valid && h.lseq >= 1u && h.lseq <= (uint32_t)F_MAXLEN;
        // pieces start phi bases before the read (its coordinates below are shifted by phi); np of them cover it
        g.phi = shortq ? phi_lane : 0u;
        g.np = shortq ? (h.lseq + g.phi + 15u) >> 4 : 1u;
        m0 = __builtin_amdgcn_readfirstlane(h.o8);
        g.row = (h.o8 - m0) * 8u;                                            // byte offset of the read's qualities in the run
        const uint32_t nch = (h.lseq + 7u) >> 3;
        // a row is read as pieces of 16 bytes: up to 16 bytes past the read's own padded bytes
        const bool fits = valid && h.o8 >= m0 && (h.o8 - m0) <= (uint32_t)(F_STAGE / 8) &&
                          g.row + 8u * nch + (shortq ? 8u : 0u) <= (uint32_t)F_STAGE;
        const unsigned long long fitmask = __ballot(fits);
        g.ntake = fitmask == ~0ull ? 64 : __builtin_ctzll(~fitmask);
        g.solo = g.ntake == 0;                                                // the first read alone is too long: general pass
        if (g.solo) g.ntake = 1;
        g.taken = lane < g.ntake && !g.solo;
        g.Tq = g.solo ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)(g.row + 8u * nch), g.ntake - 1);   // bytes of the run (scalar)
        g.fastq = g.taken && shortq;
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
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 1024; ++sl) {
            uint32_t off = (uint32_t)(sl * 1024 + lane * 16);
            off = off < lastq ? off : lastq;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qrun + off),
                                             (__attribute__((address_space(3))) void *)(stage + sl * 1024), 16, 0, 0);
        }
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
    // last until the store has reached memory.
    struct Pend { int64_t i; uint32_t slot_lo; int32_t pos, reflen; uint32_t ncig, cw0, cw1, cw2, status, flags, entry; bool simple, has; };
    Pend pend{0, 0u, 0, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, false, false};
    uint32_t gwave = 0;                                             // entries of this wave's segment of the general list
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
        // hand-over to the general pass: the wave's own segment of the list, in read order (the general kernel's
        // window follows the positions of the reads it is given)
        const unsigned long long m = __ballot(r.has);
        if (r.has) glist[(size_t)wbeg + gwave + __popcll(m & ((1ull << lane) - 1ull))] = r.entry;
        gwave += (uint32_t)__popcll(m);
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
        const uint32_t np = g.np, phi = g.phi;
        const bool solo = g.solo, taken = g.taken, fastq = g.fastq, in_ref = g.in_ref;
        const uint32_t w0 = xA.w0;
        const int32_t tabL = xA.tabL, tabR = xA.tabR;
        // ---- the wave's packed window: fold and re-anchor when the tile has moved on, or before a byte could overflow
        {
            const int32_t first_pos = __builtin_amdgcn_readfirstlane(pos);
            const int32_t want = (first_pos < 16 ? 0 : first_pos - 16) & ~15;
            if (pw_tiles >= F_FLUSH || want < pw_base || want - pw_base >= 64) {
                if (pw_tiles) fold();
                pw_base = want; pw_tiles = 0;
            }
            ++pw_tiles;
        }
        const uint32_t pw_lim = (int64_t)G - pw_base >= (int64_t)F_PW ? (uint32_t)F_PW : (uint32_t)(G > (uint32_t)pw_base ? G - (uint32_t)pw_base : 0u);
        // ---- rows: slot k of the lane holds piece (k + rot) mod np of its read (slots >= np: a copy of the last
        // piece and an index past the read, which every range test below excludes) -------------------------------
        const uint32_t rot = (uint32_t)lane % np;
        const uint8_t *qrow = rd.qual + (int64_t)o8 * 8;
        const uint8_t *srow = rd.seq + (int64_t)o8 * 4;
        const int32_t lrow = fastq ? (int32_t)g.row - (int32_t)phi : 0;        // >= -8: the pad in front of the run
        uint4 q16[F_NP];
        uint2 s8[F_NP];
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the DMA of this tile's qualities has landed, its other loads too
        F_STAMP(1);
        wave_sync();
#pragma unroll
        for (int k = 0; k < F_NP; ++k) {
            uint32_t p = (uint32_t)k + rot;
            p = p >= np ? p - np : p;
            p = (uint32_t)k < np ? p : np - 1u;
            const lds_u8 *src = stage + lrow + (int32_t)(p * 16u);
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
            const lds_u8 *src = stage + (lrow >> 1) + (int32_t)(p * 8u);
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
        // aligned-quality window [lo, hi) in PIECE coordinates (query index + phi)
        const int32_t lo = !scan ? 0 : (sc.m > 0 ? sc.a : (int32_t)lseq) + (int32_t)phi, qlen = scan ? sc.m : 0, hi = lo + qlen;
        // the 3' end's shrinking windows (A:575-576, A:637-638) need at most W-1 bytes; loaded by every lane (from
        // the start of its read when there is nothing to scan): a load under a branch is waited for at its end
        const int32_t first = !scan ? 0 : ((rev || qlen < W) ? lo : lo + qlen - W + 1) - (int32_t)phi;      // query index
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

        F_STAMP(3);          // window scan
        // ---- quality clip, results (A:589-686) ------------------------------------------------------------------
        bool general = (taken || (solo && lane == 0)) && !simple;
        bool counted = false, stored = false;
        uint32_t ncig = 0, cw[3] = {0u, 0u, 0u};
        int32_t reflen = 0;
        if (simple) {
            // the read's first quality byte (0xFF = QUAL '*') is byte phi of piece 0, which sits in slot (np - rot) mod np
            uint32_t fb = phi ? q16[0].z : q16[0].x;
#pragma unroll
            for (int k = 1; k < F_NP; ++k) fb = ((uint32_t)k + rot == np) ? (phi ? q16[k].z : q16[k].x) : fb;
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
                        const int32_t qlo = lo - (int32_t)phi, qhi = hi - (int32_t)phi;            // query indices
                        for (int32_t k = 1; k <= kmax; ++k) {
                            const uint32_t o = (uint32_t)((rev ? qlo + k - 1 : qhi - k) - tab);      // 0..15
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

        F_STAMP(4);          // quality clip, results
        // ---- counting (A:709-753 for a read without indels) -----------------------------------------------------
        uint32_t redo = 0;                        // pieces (slots) the careful loop has to do
        const int32_t qa = counted ? sc.a + (int32_t)phi : 0, qb = counted ? qa + sc.m : 0;      // piece coordinates
        if (P.do_count) {
            const int32_t dbase = ts.pos - pw_base - qa;                   // window offset of piece coordinate 0
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
                    // per base (byte): 1 = counted (quality and range), code, shift count of its counter byte
                    uint32_t f[4], cb[4], sh[4];
                    f[0] = (ok_bits4(q16[k].x, mqb) >> 7) & nibble_to_bytes(rng, 0);
                    f[1] = (ok_bits4(q16[k].y, mqb) >> 7) & nibble_to_bytes(rng, 1);
                    f[2] = (ok_bits4(q16[k].z, mqb) >> 7) & nibble_to_bytes(rng, 2);
                    f[3] = (ok_bits4(q16[k].w, mqb) >> 7) & nibble_to_bytes(rng, 3);
                    spread_codes(s8[k], cb);
                    uint32_t bad = 0;
#pragma unroll
                    for (int d = 0; d < 4; ++d) { bad |= (not_acgt(cb[d]) >> 7) & f[d]; sh[d] = shift_bytes(cb[d]); }
                    // (a counted N is rare -- N calls come with low qualities -- and takes the careful loop too)
                    const bool safe = bad == 0u && pw_lim >= 16u && (uint32_t)d0 <= pw_lim - 16u;
                    if (safe) {
                        lds_u8 *const wb = wrep + (uint32_t)d0 * 4u;
                        lds_add((lds_u32 *)(wb + 0), shl_byte<0>(sh[0], f[0]));   lds_add((lds_u32 *)(wb + 4), shl_byte<1>(sh[0], f[0]));
                        lds_add((lds_u32 *)(wb + 8), shl_byte<2>(sh[0], f[0]));   lds_add((lds_u32 *)(wb + 12), shl_byte<3>(sh[0], f[0]));
                        lds_add((lds_u32 *)(wb + 16), shl_byte<0>(sh[1], f[1]));  lds_add((lds_u32 *)(wb + 20), shl_byte<1>(sh[1], f[1]));
                        lds_add((lds_u32 *)(wb + 24), shl_byte<2>(sh[1], f[1]));  lds_add((lds_u32 *)(wb + 28), shl_byte<3>(sh[1], f[1]));
                        lds_add((lds_u32 *)(wb + 32), shl_byte<0>(sh[2], f[2]));  lds_add((lds_u32 *)(wb + 36), shl_byte<1>(sh[2], f[2]));
                        lds_add((lds_u32 *)(wb + 40), shl_byte<2>(sh[2], f[2]));  lds_add((lds_u32 *)(wb + 44), shl_byte<3>(sh[2], f[2]));
                        lds_add((lds_u32 *)(wb + 48), shl_byte<0>(sh[3], f[3]));  lds_add((lds_u32 *)(wb + 52), shl_byte<1>(sh[3], f[3]));
                        lds_add((lds_u32 *)(wb + 56), shl_byte<2>(sh[3], f[3]));  lds_add((lds_u32 *)(wb + 60), shl_byte<3>(sh[3], f[3]));
                    } else {
                        redo |= 1u << k;
                    }
                }
            }
        }
        F_STAMP(5);          // counting
        if (__ballot(redo != 0u)) {
            // careful loop (rare): bases of the flagged pieces one by one, straight from memory into the 32-bit counters
            bool bad = false;
            if (redo) {
                const int32_t qaq = sc.a, qbq = sc.a + sc.m;               // query indices
                for (int k = 0; k < F_NP; ++k) {
                    if (!((redo >> k) & 1u)) continue;
                    uint32_t p = (uint32_t)k + rot;
                    p = p >= np ? p - np : p;
                    const int32_t j0 = (int32_t)(p * 16u) - (int32_t)phi;
                    for (int32_t q = j0 < qaq ? qaq : j0; q < j0 + 16 && q < qbq; ++q) {
                        if ((int32_t)qrow[q] < mq) continue;
                        const uint32_t sb = srow[q >> 1];
                        const uint32_t col = col_of_code((q & 1) ? (sb & 15u) : (sb >> 4));
                        const int32_t rp = ts.pos + (q - qaq);
                        const uint32_t d = (uint32_t)(rp - bw_base);
                        if (col > 4u || (uint32_t)rp >= G) bad = true;
                        else if (d < (uint32_t)F_BW && col < (uint32_t)F_NPL) lds_add(bwin + col * F_BW + d, 1u);
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
        F_STAMP(6);          // careful loop
        hA = hB; gA = gB; xA = xB; m0A = m0B; hB = hC;
        i0 = i1; i1 = i2;
    }
    store_pending(pend);
    if (pw_tiles && wbeg < wend) fold();

    __syncthreads();
    for (int i = tid; i < F_NPL * F_BW; i += F_WAVES * 64) {
        const uint32_t v = bwin[i];
        if (v) {
            const int sym = i / F_BW, d = i - sym * F_BW;
            if ((uint32_t)(bw_base + d) < G) atomicAdd(&counts[(size_t)(bw_base + d) * AMP_NSYM + sym], v);
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
    F_STAMP_OUT;
    if (lane == 0) gcnt[blockIdx.x * F_WAVES + wave] = gwave;
}

// Dense list of the reads the fast kernel handed over + the geometry of the general pass.  Block b places the
// segment of fast-kernel wave b behind the totals of the waves before it (the per-wave counts are a few KB in L2);
// read order is kept.
__global__ void __launch_bounds__(256)
k_gcompact(const uint32_t *__restrict__ glist, const uint32_t *__restrict__ gcnt, int reads_per_block, int64_t n_reads,
           uint32_t *__restrict__ dense, GenGeo *geo, uint32_t gen_grid) {
    __shared__ uint32_t s_part[4];
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) acc += gcnt[b];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    const uint32_t off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const uint32_t cnt = gcnt[blockIdx.x];
    const int per_wave = reads_per_block / F_WAVES;
    const int64_t wbeg = (int64_t)(blockIdx.x / F_WAVES) * reads_per_block + (int64_t)(blockIdx.x % F_WAVES) * per_wave;
    const uint32_t *src = glist + (wbeg < n_reads ? wbeg : 0);
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
