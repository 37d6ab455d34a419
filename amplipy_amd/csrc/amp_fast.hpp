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
#include "amp_bf.hpp"

namespace amp {

#ifndef AMP_F_WAVES
#define AMP_F_WAVES 8
#endif
#ifndef AMP_F_LDSPAD
#define AMP_F_LDSPAD 0
#endif
constexpr int F_WAVES = AMP_F_WAVES;  // waves per block (one block per CU: LDS)
#ifndef AMP_F4_LEAN
#define AMP_F4_LEAN 1    // counting from 16 good-quality bits per piece with the lean piece code below (count_piece5); 0: round 2's fast_count_piece
#endif
#ifndef AMP_F4_BF
#define AMP_F4_BF 1      // 1: the trims of a read in their branch-free form (amp_bf.hpp, fuzzed against the branchy forms on the CPU; 1 % faster); 0: the branchy closed forms of amp_read.hpp
#endif
__device__ __forceinline__ Bf bf_of(const Cig2 &s) { return Bf{s.a, s.m1, s.k, s.m2, s.c, s.kind, s.op, s.punt ? 1u : 0u}; }
__device__ __forceinline__ Cig2 cig2_of(const Bf &b) { return Cig2{b.op, b.a, b.m1, b.k, b.m2, b.c, b.kind, b.punt != 0u}; }
constexpr int F_NP = 10;              // 16-base pieces per read held in registers
constexpr int F_MAXLEN = 152;         // longest read the fast path takes: F_NP pieces must cover it from 8 bases before its start
constexpr int F_PW = 256;             // reference positions covered by a wave's packed window
#ifndef AMP_F_REPS
#define AMP_F_REPS 8
#endif
constexpr int F_REP = AMP_F_REPS;     // replicas of the packed window
#ifndef AMP_F_ADD64
#define AMP_F_ADD64 1                 // 1: eight 64-bit adds per piece into even / odd arrays (count_piece5q), F_REP / 2 replicas of two arrays each; 0: sixteen 32-bit adds
#endif
constexpr int F_REPW = F_PW + 1;      // words per replica: one word of skew, so that replica r is shifted by r banks
#ifndef AMP_F_BW
#define AMP_F_BW 512
#endif
constexpr int F_BW = AMP_F_BW;        // reference positions covered by the block's 32-bit window
constexpr int F_NPL = 4;              // its base planes: A C G T (a counted N goes straight to the table)
constexpr int F_BPL = 6;              // ... followed by '-' (deletions) and the insertion-event tally
constexpr int F_MAXINS = 8;           // longest insertion / deletion of a read the fast path takes
constexpr int F_MAXDEL = 16;
constexpr uint32_t F_EVGRAN = 64;     // event-list slots a wave reserves at a time
constexpr int F_STAGE = 10240;        // bytes of a wave's staging buffer = the longest run of quality bytes a tile may span
constexpr int F_PAD = 16;             // bytes in front of the staged run (rows that start 8 bases early)
constexpr int F_FLUSH = 15;           // tiles between two folds of a packed window (16 increments per counter and tile at most)

struct FastGrid { int64_t grid, rpb; };
static inline FastGrid fast_grid(int64_t n_reads, int n_cu) {
#ifndef AMP_F_BPC
#define AMP_F_BPC 1
#endif
    // AMP_F_BPC blocks per CU (one is resident): a block owns a contiguous range of whole tiles of 64 reads, which its
    // waves take one by one (at least two tiles per wave)
    int64_t rpb = (n_reads + AMP_F_BPC * (int64_t)n_cu - 1) / (AMP_F_BPC * (int64_t)n_cu);
    rpb = ((rpb + 63) / 64) * 64;
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

// LDS adds of the counting phase, written as inline assembly ON PURPOSE.  While a tile is counted the next tile's
// quality bytes are on their way into the staging buffer by LDS-DMA, and the compiler answers every LDS access it can
// see with s_waitcnt vmcnt(0) as long as such a load is in flight (it cannot prove that the counters and the staging
// buffer are different memory): the counting phase would wait for the bytes it is meant to overlap with.  An
// instruction inside an asm statement is not tracked.  LDS operations of a wave complete in order, so the waits the
// compiler places for its own LDS reads stay sufficient with these adds in between.
__device__ __forceinline__ void lds_add_nt(lds_u32 *p, uint32_t v) {
    asm volatile("ds_add_u32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
// counter word at wb + 4 * B  +=  (byte J of val) << (byte J of sh): the SDWA shift and the add of one base
template <int J, int B>
__device__ __forceinline__ void add_base(uint32_t wb, uint32_t sh, uint32_t val) {
    uint32_t t;
    if (J == 0) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(val), "v"(wb), "n"(4 * B) : "memory");
    if (J == 1) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(val), "v"(wb), "n"(4 * B) : "memory");
    if (J == 2) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(val), "v"(wb), "n"(4 * B) : "memory");
    if (J == 3) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(val), "v"(wb), "n"(4 * B) : "memory");
}

// 16 failing-window bits of one piece: bit b set <=> the W-byte window starting at byte b of q (continued in nx) sums to < thr
template <int W>
__device__ __forceinline__ uint32_t piece_fail_bits(const uint4 &q, const uint2 &nx, uint32_t thr) {
    if (W == 4) {
        // v_qsad_pk_u16_u8 gives four sliding 4-byte sums per instruction and ADDS its third operand to each: with
        // 65536 - thr there, bit 15 of a sum is "sum < thr" (thr <= 2048).  The 16 sign bits are gathered by a packed
        // shift (bit 15 -> bit 0 of each half) and v_dot4_u32_u8 with power-of-two weights.
        typedef unsigned short amp_u16x2 __attribute__((ext_vector_type(2)));
        const uint64_t nthr = (uint64_t)((0x10000u - thr) & 0xFFFFu) * 0x0001000100010001ull;
        const uint64_t s0 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.x | ((uint64_t)q.y << 32), 0u, nthr);
        const uint64_t s1 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.y | ((uint64_t)q.z << 32), 0u, nthr);
        const uint64_t s2 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.z | ((uint64_t)q.w << 32), 0u, nthr);
        const uint64_t s3 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.w | ((uint64_t)nx.x << 32), 0u, nthr);
        const uint32_t d[8] = {(uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)s1, (uint32_t)(s1 >> 32),
                               (uint32_t)s2, (uint32_t)(s2 >> 32), (uint32_t)s3, (uint32_t)(s3 >> 32)};
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = (1u << (2 * j)) | (2u << (2 * j + 16));          // weights of the two windows of a dword
            const uint32_t a = __builtin_bit_cast(uint32_t, __builtin_bit_cast(amp_u16x2, d[j]) >> (unsigned short)15);
            const uint32_t b = __builtin_bit_cast(uint32_t, __builtin_bit_cast(amp_u16x2, d[j + 4]) >> (unsigned short)15);
            lo = __builtin_amdgcn_udot4(a, w, lo, false);
            hi = __builtin_amdgcn_udot4(b, w, hi, false);
        }
        return lo | (hi << 8);
    }
    return window_fail_bits16<W>(make_uint2(q.x, q.y), make_uint2(q.z, q.w), thr) |
           (window_fail_bits16<W>(make_uint2(q.z, q.w), nx, thr) << 8);
}

// zero when every one of the 8 base codes packed in x is one of A C G T (exactly one bit per nibble)
__device__ __forceinline__ uint32_t acgt_defect8(uint32_t x) {
    const uint32_t m = 0x11111111u;
    return ((x & m) + ((x >> 1) & m) + ((x >> 2) & m) + ((x >> 3) & m)) ^ m;
}

// per byte: 8 * plane number of an A C G T code (a byte permute: 1 2 4 8 -> selector 1 2 4 0 -> 0 8 16 24)
__device__ __forceinline__ uint32_t shift_bytes_acgt(uint32_t cb) {
    return __builtin_amdgcn_perm(0x00000010u, 0x00080018u, cb & 0x07070707u);
}

// One piece (16 bases in registers: qualities q, packed codes sq) against the counted query range [qa_, qb_)
// (piece coordinates, the piece starts at j0): every base that is inside the range and good enough adds 1 to
// its counter byte.  dbase_ = window offset of piece coordinate 0, wrep = LDS address of
// the lane's replica of the wave's packed window, pw_lim = its usable width.  Returns true when the piece has to be redone
// by the careful loop (a counted code outside A C G T, or the piece leaves the packed window).
__device__ __forceinline__ bool fast_count_piece(const uint4 &q, const uint2 &sq, int32_t j0, int32_t qa_, int32_t qb_, int32_t dbase_,
                                             uint32_t mqb, uint32_t pw_lim, uint32_t wrep) {
    // Straight-line code on purpose: the slots are rotated per lane, so some lane of the wave has work in every
    // slot and a branch around an empty piece would never be taken by the whole wave; what a divergent branch costs
    // here is its v_cmp -> s_and_saveexec -> s_cbranch chain, which two waves per SIMD cannot hide.  A lane without
    // counted bases adds zeros.
    int32_t klo = qa_ - j0, khi = qb_ - j0;
    klo = klo < 0 ? 0 : (klo > 16 ? 16 : klo); khi = khi > 16 ? 16 : (khi < 0 ? 0 : khi);
    const int32_t d0 = dbase_ + j0;                                 // window offset of the piece's base 0
    const uint32_t rng = ((1u << khi) - 1u) & ~((1u << klo) - 1u);  // empty when khi <= klo
    // per base (byte): 1 = counted (quality and range), code, shift count of its counter byte
    uint32_t f[4], cb[4], sh[4];
    f[0] = (ok_bits4(q.x, mqb) >> 7) & nibble_to_bytes(rng, 0);
    f[1] = (ok_bits4(q.y, mqb) >> 7) & nibble_to_bytes(rng, 1);
    f[2] = (ok_bits4(q.z, mqb) >> 7) & nibble_to_bytes(rng, 2);
    f[3] = (ok_bits4(q.w, mqb) >> 7) & nibble_to_bytes(rng, 3);
    spread_codes(sq, cb);
#pragma unroll
    for (int d = 0; d < 4; ++d) sh[d] = shift_bytes_acgt(cb[d]);
    const uint32_t counted_any = f[0] | f[1] | f[2] | f[3];
    const bool inwin = pw_lim >= 16u && (uint32_t)d0 <= pw_lim - 16u;
    bool redo = counted_any != 0u && !inwin;                        // the piece leaves the packed window
    if (acgt_defect8(sq.x) | acgt_defect8(sq.y)) {                  // rare: some code of the piece is not A C G T
        uint32_t bad = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) bad |= (not_acgt(cb[d]) >> 7) & f[d];
        redo = redo || bad != 0u;                                   // ... and it is a counted one (a counted N takes the careful loop too)
    }
    const uint32_t keep = redo ? 0u : 0xFFFFFFFFu;
    const uint32_t wb = wrep + (inwin ? (uint32_t)d0 * 4u : 0u);
#if defined(AMP_DEV) && defined(AMP_ABL)
    if (AMP_ABL & 1) { asm volatile("" : : "v"(wb), "v"(sh[0] + sh[1] + sh[2] + sh[3]), "v"((f[0] + f[1] + f[2] + f[3]) & keep)); return redo; }   // ablation: no adds
#endif
    const uint32_t f0 = f[0] & keep, f1 = f[1] & keep, f2 = f[2] & keep, f3 = f[3] & keep;
    add_base<0, 0>(wb, sh[0], f0);   add_base<1, 1>(wb, sh[0], f0);   add_base<2, 2>(wb, sh[0], f0);   add_base<3, 3>(wb, sh[0], f0);
    add_base<0, 4>(wb, sh[1], f1);   add_base<1, 5>(wb, sh[1], f1);   add_base<2, 6>(wb, sh[1], f1);   add_base<3, 7>(wb, sh[1], f1);
    add_base<0, 8>(wb, sh[2], f2);   add_base<1, 9>(wb, sh[2], f2);   add_base<2, 10>(wb, sh[2], f2);  add_base<3, 11>(wb, sh[2], f2);
    add_base<0, 12>(wb, sh[3], f3);  add_base<1, 13>(wb, sh[3], f3);  add_base<2, 14>(wb, sh[3], f3);  add_base<3, 15>(wb, sh[3], f3);
    return redo;
}

// ---- the lean piece code (shared with the second-generation kernel, amp_fast5.hpp) ------------------------------------------
// bit 7 of every byte: quality >= mq (mq <= 128, mqb = mq in every byte); three instructions
__device__ __forceinline__ uint32_t ok80(uint32_t q, uint32_t mqb) {
    const uint32_t t = ((q & 0x7F7F7F7Fu) | 0x80808080u) - mqb;
    return (t | q) & 0x80808080u;
}
// 16 bits "quality >= mq" of a piece
__device__ __forceinline__ uint32_t ok_bits16(const uint4 &q, uint32_t mqb) {
    // byte flags 0x80 times the weights 1 2 4 8 (16 32 64 128) add up to the nibble << 7
    uint32_t lo = __builtin_amdgcn_udot4(ok80(q.x, mqb), 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(ok80(q.y, mqb), 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(ok80(q.z, mqb), 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(ok80(q.w, mqb), 0x80402010u, hi, false);
    return (lo >> 7) | ((hi >> 7) << 8);
}
// zero when all 8 nibbles of x hold exactly one bit (A C G T)
__device__ __forceinline__ uint32_t nibbles_bad(uint32_t x, uint32_t y) {
    const uint32_t z = ((x - 0x11111111u) & ~x & 0x88888888u) | ((y - 0x11111111u) & ~y & 0x88888888u);      // a zero nibble
    return z | ((uint32_t)(__builtin_popcount(x) + __builtin_popcount(y)) ^ 16u);
}
// counter word at wb + 4 * B  +=  (byte JV of val) << (byte JS of sh)
template <int JS, int JV, int B>
__device__ __forceinline__ void add_base2(uint32_t wb, uint32_t sh, uint32_t val) {
#define AMP_AB2(js, jv) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #js " src1_sel:BYTE_" #jv "\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(val), "v"(wb), "n"(4 * B) : "memory")
    uint32_t t;
    if (JS == 0 && JV == 0) AMP_AB2(0, 0); if (JS == 0 && JV == 1) AMP_AB2(0, 1); if (JS == 0 && JV == 2) AMP_AB2(0, 2); if (JS == 0 && JV == 3) AMP_AB2(0, 3);
    if (JS == 1 && JV == 0) AMP_AB2(1, 0); if (JS == 1 && JV == 1) AMP_AB2(1, 1); if (JS == 1 && JV == 2) AMP_AB2(1, 2); if (JS == 1 && JV == 3) AMP_AB2(1, 3);
    if (JS == 2 && JV == 0) AMP_AB2(2, 0); if (JS == 2 && JV == 1) AMP_AB2(2, 1); if (JS == 2 && JV == 2) AMP_AB2(2, 2); if (JS == 2 && JV == 3) AMP_AB2(2, 3);
    if (JS == 3 && JV == 0) AMP_AB2(3, 0); if (JS == 3 && JV == 1) AMP_AB2(3, 1); if (JS == 3 && JV == 2) AMP_AB2(3, 2); if (JS == 3 && JV == 3) AMP_AB2(3, 3);
#undef AMP_AB2
}
// One piece (16 bases): packed codes sq, counted-base bits m16 (bit b: base b of the piece is inside the counted range
// and good enough), window offset d0 of its base 0, against the lane's replica of the wave's packed window.
// Returns true when the careful loop has to redo the piece (a code outside A C G T in it, or it leaves the window).
__device__ __forceinline__ uint32_t count_piece5(const uint2 &sq, uint32_t m16, int32_t d0, int32_t lim16, uint32_t wrep) {
    // (bit operations on purpose: with && / || the compiler branches around the tests)
    const bool inwin = (d0 >= 0) & (d0 <= lim16);                       // lim16 = the window's width - 16 (negative: nothing fits)
    const bool redo = (m16 != 0u) & (!inwin | (nibbles_bad(sq.x, sq.y) != 0u));
    const uint32_t m = redo ? 0u : m16;
    // shift counts of the even / odd bases of each half (the high nibble of a byte is the even base)
    const uint32_t se0 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, (sq.x >> 4) & 0x07070707u);
    const uint32_t so0 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, sq.x & 0x07070707u);
    const uint32_t se1 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, (sq.y >> 4) & 0x07070707u);
    const uint32_t so1 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, sq.y & 0x07070707u);
    const uint32_t f0 = nibble_to_bytes(m, 0), f1 = nibble_to_bytes(m, 1), f2 = nibble_to_bytes(m, 2), f3 = nibble_to_bytes(m, 3);
    const uint32_t wb = wrep + (inwin ? (uint32_t)d0 * 4u : 0u);
    // the 16 shift + add pairs of the piece in ONE asm statement (round 4: between separate statements the compiler puts hazard
    // s_nops it cannot rule out -- ~100 issue slots per tile); two temporaries alternate, a shift is issued while the add before it
    // is on its way.  Base b of the piece: shift count = byte (b >> 1) & 3 of the even / odd shift word of its half, value = byte
    // b & 3 of the flag word of its group of four
#define F_SH(d, sh, js, v, jv) "v_lshlrev_b32_sdwa " d ", " sh ", " v " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #js " src1_sel:BYTE_" #jv "\n\t"
#define F_AD(t, off) "ds_add_u32 %2, " t " offset:" #off "\n\t"
    {
        uint32_t t0, t1;
        asm volatile(F_SH("%0", "%3", 0, "%7", 0) F_SH("%1", "%4", 0, "%7", 1) F_AD("%0", 0) F_SH("%0", "%3", 1, "%7", 2) F_AD("%1", 4) F_SH("%1", "%4", 1, "%7", 3)
                     F_AD("%0", 8) F_SH("%0", "%3", 2, "%8", 0) F_AD("%1", 12) F_SH("%1", "%4", 2, "%8", 1) F_AD("%0", 16) F_SH("%0", "%3", 3, "%8", 2)
                     F_AD("%1", 20) F_SH("%1", "%4", 3, "%8", 3) F_AD("%0", 24) F_SH("%0", "%5", 0, "%9", 0) F_AD("%1", 28) F_SH("%1", "%6", 0, "%9", 1)
                     F_AD("%0", 32) F_SH("%0", "%5", 1, "%9", 2) F_AD("%1", 36) F_SH("%1", "%6", 1, "%9", 3) F_AD("%0", 40) F_SH("%0", "%5", 2, "%10", 0)
                     F_AD("%1", 44) F_SH("%1", "%6", 2, "%10", 1) F_AD("%0", 48) F_SH("%0", "%5", 3, "%10", 2) F_AD("%1", 52) F_SH("%1", "%6", 3, "%10", 3)
                     F_AD("%0", 56) "ds_add_u32 %2, %1 offset:60"
                     : "=&v"(t0), "=&v"(t1) : "v"(wb), "v"(se0), "v"(so0), "v"(se1), "v"(so1), "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "memory");
    }
#undef F_SH
#undef F_AD
    return redo ? 1u : 0u;
}
// The same with EIGHT 64-bit adds instead of sixteen 32-bit ones: an LDS atomic is issued per wave-instruction whatever its width,
// and the counting phase is bound by how fast a SIMD gets its atomics into the LDS (sixteen cycles apiece when the four SIMDs of a
// CU share the pipe fairly), not by its vector instructions.  ds_add_u64 faults on an address that is not a multiple of 8, and
// whether a piece starts on an even or an odd window offset depends on the read: a replica therefore holds TWO arrays, the second
// ooff bytes behind the first with ooff = 4 (mod 8) (an odd number of words per array), and a piece with an odd offset adds into the
// second one, where ITS pairs are aligned.  The two halves of a pair come from two SDWA shifts into the FIXED registers v[252:253] / v[254:255] (an asm operand
// cannot name the halves of a 64-bit register).  Counter bytes never carry (flush discipline), so the low word cannot spill over.
__device__ __forceinline__ uint32_t count_piece5q(const uint2 &sq, uint32_t m16, int32_t d0, int32_t lim16, uint32_t wrep, uint32_t ooff) {
    const bool inwin = (d0 >= 0) & (d0 <= lim16);
    const bool redo = (m16 != 0u) & (!inwin | (nibbles_bad(sq.x, sq.y) != 0u));
    const uint32_t m = redo ? 0u : m16;
    const uint32_t se0 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, (sq.x >> 4) & 0x07070707u);
    const uint32_t so0 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, sq.x & 0x07070707u);
    const uint32_t se1 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, (sq.y >> 4) & 0x07070707u);
    const uint32_t so1 = __builtin_amdgcn_perm(0x00000010u, 0x00080018u, sq.y & 0x07070707u);
    const uint32_t f0 = nibble_to_bytes(m, 0), f1 = nibble_to_bytes(m, 1), f2 = nibble_to_bytes(m, 2), f3 = nibble_to_bytes(m, 3);
    const uint32_t wb = wrep + (inwin ? (uint32_t)d0 * 4u + (((uint32_t)d0 & 1u) ? ooff : 0u) : 0u);
#define F_SH(d, sh, js, v, jv) "v_lshlrev_b32_sdwa " d ", " sh ", " v " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #js " src1_sel:BYTE_" #jv "\n\t"
#define F_AQ(t, off) "ds_add_u64 %0, " t " offset:" #off "\n\t"
    asm volatile(F_SH("v252", "%1", 0, "%5", 0) F_SH("v253", "%2", 0, "%5", 1) F_AQ("v[252:253]", 0)
                 F_SH("v254", "%1", 1, "%5", 2) F_SH("v255", "%2", 1, "%5", 3) F_AQ("v[254:255]", 8)
                 F_SH("v252", "%1", 2, "%6", 0) F_SH("v253", "%2", 2, "%6", 1) F_AQ("v[252:253]", 16)
                 F_SH("v254", "%1", 3, "%6", 2) F_SH("v255", "%2", 3, "%6", 3) F_AQ("v[254:255]", 24)
                 F_SH("v252", "%3", 0, "%7", 0) F_SH("v253", "%4", 0, "%7", 1) F_AQ("v[252:253]", 32)
                 F_SH("v254", "%3", 1, "%7", 2) F_SH("v255", "%4", 1, "%7", 3) F_AQ("v[254:255]", 40)
                 F_SH("v252", "%3", 2, "%8", 0) F_SH("v253", "%4", 2, "%8", 1) F_AQ("v[252:253]", 48)
                 F_SH("v254", "%3", 3, "%8", 2) F_SH("v255", "%4", 3, "%8", 3) "ds_add_u64 %0, v[254:255] offset:56"
                 : : "v"(wb), "v"(se0), "v"(so0), "v"(se1), "v"(so1), "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "memory", "v252", "v253", "v254", "v255");
#undef F_SH
#undef F_AQ
    return redo ? 1u : 0u;
}
// bits [klo, khi) of a 16-bit mask, klo / khi clamped to 0..16
__device__ __forceinline__ uint32_t range_bits16(int32_t klo, int32_t khi) {
    klo = klo < 0 ? 0 : (klo > 16 ? 16 : klo); khi = khi > 16 ? 16 : (khi < 0 ? 0 : khi);
    return ((1u << khi) - 1u) & ~((1u << klo) - 1u);                    // empty when khi <= klo
}

// in-kernel phase stamps (development builds only; the numbers are shares, not durations); AMP_ABL = ablation
// builds (parts of the kernel switched off to time the rest: results are wrong on purpose), without stamps
#if defined(AMP_DEV) && (!defined(AMP_ABL) || defined(AMP_ABL_STAMPS))
#define F_DBG_PARAM , uint32_t *f_dbg
#define F_GLOB ++f_glob
#define F_EPIW(k) do { __builtin_amdgcn_s_waitcnt(0x4F74); f_epw[0] = wall_clock64(); __builtin_amdgcn_s_waitcnt(0x0F7A); f_epw[1] = wall_clock64(); __builtin_amdgcn_s_waitcnt(0); f_ep[k] = wall_clock64(); } while (0)
#define F_EPI(k) do { __builtin_amdgcn_s_waitcnt(0); f_ep[k] = wall_clock64(); } while (0)
#define F_DBG_ARG(x) , (x)
#define F_STAMP_DECL unsigned long long f_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f_prev = __builtin_amdgcn_s_memtime(); const unsigned long long f_k0 = f_prev, f_w0 = wall_clock64(); unsigned long long f_ep[3] = {0, 0, 0}, f_epw[2] = {0, 0}; uint32_t f_glob = 0
#define F_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); unsigned long long f_n = __builtin_amdgcn_s_memtime(); f_t[k] += f_n - f_prev; f_prev = f_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define F_STAMP_VM(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0x0070); unsigned long long f_n = __builtin_amdgcn_s_memtime(); f_t[k] += f_n - f_prev; f_prev = f_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define F_STAMP_OUT do { f_t[0] = __builtin_amdgcn_s_memtime() - f_k0; const unsigned long long f_w1 = wall_clock64(); f_t[7] = f_w1 - f_w0; if (lane == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&ctr[8 + k], f_t[k]); \
    atomicMax(&ctr[4], f_t[7]); atomicMax(&ctr[5], ~f_t[7]); atomicMax(&ctr[6], ~f_w0); atomicMax(&ctr[7], f_w1); \
    if (f_dbg && wave == 0) { uint32_t *f_o = f_dbg + blockIdx.x * 8; f_o[0] = (uint32_t)f_t[7]; f_o[1] = (uint32_t)f_w0; for (int k = 1; k < 7; ++k) f_o[1 + k] = (uint32_t)(f_t[k] >> 4); } \
    if (f_dbg) { uint32_t *f_o = f_dbg + 2048 + (blockIdx.x * F_WAVES + wave) * 6; f_o[0] = (uint32_t)((f_t[1] + f_t[2] + f_t[3] + f_t[4] + f_t[5] + f_t[6]) >> 4); f_o[1] = (uint32_t)(f_epw[0] - f_w0); f_o[2] = (uint32_t)(f_epw[1] - f_epw[0]); f_o[3] = (uint32_t)(f_ep[0] - f_epw[1]); f_o[4] = (uint32_t)(f_ep[1] - f_ep[0]); f_o[5] = (uint32_t)(f_ep[2] - f_ep[1]); } } } while (0)
#else
#define F_DBG_PARAM
#define F_DBG_ARG(x)
#define F_GLOB
#define F_EPIW(k)
#define F_EPI(k)
#define F_STAMP_DECL
#define F_STAMP(k)
#define F_STAMP_VM(k)
#define F_STAMP_OUT
#endif

// The kernel's arguments are passed one by one, every pointer followed by a 32-bit value (round 4): a struct argument -- and a run of
// adjacent pointers -- is loaded as ONE wide register tuple, and a kernel as short of scalar registers as this one spills and
// reloads the whole tuple around every use of one field.
#define F_ARGS \
    const int32_t *a_pos, int32_t a_min_quality, const uint16_t *a_flag, int32_t a_window, const int32_t *a_tlen, int32_t a_do_trim, \
    const uint32_t *a_lseq, int32_t a_do_count, const uint32_t *a_cig_off32, int32_t a_ref_len, const uint32_t *a_cig, int32_t a_max_primer_len, \
    const uint32_t *a_seq_off8, int32_t reads_per_block, const uint8_t *a_seq, int32_t a_epoch, const uint8_t *a_qual, int32_t pad1, \
    const int32_t *a_min_start, int32_t pad2, const int32_t *a_max_end, int32_t pad3, \
    int32_t *a_new_pos, int32_t pad4, uint32_t *a_new_ncig, int32_t pad5, uint32_t *a_new_cig, int32_t pad6, int32_t *a_o_ref_len, int32_t pad7, \
    uint8_t *a_trim_flags, int32_t pad8, uint8_t *a_status, int32_t pad9, \
    uint32_t *counts, int32_t pad10, amp_ins_event *a_ev, int32_t pad11, unsigned long long *a_ctr, int32_t pad12, uint32_t *a_ins_at, int32_t pad13, \
    uint32_t *glist, int32_t pad14, uint32_t *gcnt, int32_t pad15, \
    int64_t a_n_reads, int32_t pad17, uint64_t read_base, int32_t pad18, long long a_ev_cap
#define F_ARGS_PASS(P, rd, out, eb, rpb) \
    (rd).pos, (P).min_quality, (rd).flag, (P).window, (rd).tlen, (P).do_trim, (rd).lseq, (P).do_count, (rd).cig_off32, (P).ref_len, (rd).cig, \
    (P).max_primer_len, (rd).seq_off8, (rpb), (rd).seq, (int32_t)(P).epoch, (rd).qual, 0, (P).min_start, 0, (P).max_end, 0, (out).new_pos, 0, (out).new_ncig, 0, \
    (out).new_cig, 0, (out).ref_len, 0, (out).trim_flags, 0, (out).status, 0, counts, 0, (eb).ev, 0, (eb).ctr, 0, (eb).ins_at, 0, glist, 0, gcnt, 0, \
    (rd).n_reads, 0, read_base, 0, (eb).cap

#if AMP_F_ADD64
#define F_COUNT5(sq, m, d0, lim, wr) count_piece5q(sq, m, d0, lim, wr, (uint32_t)(F_REPW * 4))
#else
#define F_COUNT5(sq, m, d0, lim, wr) count_piece5(sq, m, d0, lim, wr)
#endif
template <int W>
__global__ void __launch_bounds__(F_WAVES * 64, 2)
k_fast(F_ARGS F_DBG_PARAM) {
    const KParams P{a_min_quality, a_window, a_do_trim, a_do_count, a_ref_len, a_max_primer_len, a_min_start, a_max_end, (uint32_t)a_epoch};
    const amp_dev_reads rd{a_n_reads, a_pos, a_flag, a_tlen, a_lseq, a_cig_off32, a_cig, a_seq_off8, a_seq, a_qual, 0, 0};
    const DevOut out{a_new_pos, a_new_ncig, a_new_cig, a_o_ref_len, a_trim_flags, a_status};
    const EventBuf eb{a_ev, a_ctr, a_ins_at, a_ev_cap};
    // THREE separate LDS objects, not one struct: the compiler orders every LDS access behind LDS-DMA loads in flight
    // (s_waitcnt vmcnt) unless alias scopes tell it that the access cannot touch the DMA's destination, and it only
    // builds those scopes per LDS variable.  With one struct every counter add waited for the next tile's bytes.
    __shared__ uint4 s_stage[F_WAVES][(F_PAD + F_STAGE + 16) / 16];   // per wave: the tile's quality bytes, then its packed bases
    __shared__ __attribute__((aligned(8))) uint32_t s_pwin[F_WAVES][F_REP * F_REPW];      // per wave: packed counters, byte c of a word = base c (A C G T)
    __shared__ uint32_t s_bwin[F_BPL * F_BW];                         // the block's window, 32-bit counters
    __shared__ uint32_t s_ticket, s_gcur;                             // next tile of the block to hand out; entries of its general list
#if AMP_F_LDSPAD > 0
    __shared__ uint32_t s_pad[AMP_F_LDSPAD / 4];                      // (occupancy experiments: keeps a second block off the CU)
    if (P.ref_len == -12345) s_pad[threadIdx.x] = 1;
#endif
    unsigned long long *const ctr = eb.ctr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    F_STAMP_DECL;
    const int64_t n = rd.n_reads;
    const int64_t rb = (int64_t)blockIdx.x * reads_per_block;
    const int64_t re = rb + reads_per_block < n ? rb + reads_per_block : n;
    lds_u32 *const bwin = (lds_u32 *)s_bwin;
    lds_u32 *const pwin = (lds_u32 *)s_pwin[wave];
    for (int i = tid; i < F_BPL * F_BW; i += F_WAVES * 64) bwin[i] = 0;
    for (int i = lane; i < F_REP * F_REPW; i += 64) pwin[i] = 0;
    if (lane < F_PAD / 4) ((lds_u32 *)s_stage[wave])[lane] = 0x11111111u;      // (the bytes in front of a staged run are read as bases by the lanes whose pieces start 8 bases early)
    if (tid == 0) { s_ticket = 0; s_gcur = 0; }
    if (tid == 0 && blockIdx.x == 0) { eb.ctr[26] = 0ull; eb.ctr[27] = 0ull; eb.ctr[28] = 0ull; }      // k_gcompact / k_long's counters (amp_wave.hpp)
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
    // Tiles are handed out one by one (a counter in LDS): the waves of a block do not run at the same speed -- of the two
    // waves that share a SIMD the older one gets most of the issue slots -- and with equal shares the CU would idle while
    // the slower half finishes.  A wave holds three tickets: the tile it computes, the tile whose bytes are on their way
    // and the tile whose header is being read.
    const uint32_t n_tb = re > rb ? (uint32_t)((re - rb + 63) / 64) : 0u;
    auto take_ticket = [&]() {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((lds_u32 *)&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    const int64_t wend = re;                                        // (a tile's lanes past the block's reads are idle)
    unsigned long long n_err = 0;
    // lane constants of the bank plan (see the head of this file)
#ifndef AMP_F_REPSHIFT
#define AMP_F_REPSHIFT 2
#endif
    const uint32_t rep = (F_REP & (F_REP - 1)) ? ((uint32_t)lane >> AMP_F_REPSHIFT) % (uint32_t)F_REP : ((uint32_t)lane >> AMP_F_REPSHIFT) & (uint32_t)(F_REP - 1);
    const uint32_t phi_lane = ((uint32_t)lane >> 1) & 1u ? 8u : 0u;
#if AMP_F_ADD64
    // (replica r = arrays 2r -- pieces that start on an even window offset -- and 2r + 1 -- odd ones; F_REPW is odd, so the second array
    //  starts 4 bytes off an 8-byte boundary and word p of it is 8-byte aligned for odd p)
    lds_u8 *const wrep = (lds_u8 *)pwin + (rep & (uint32_t)(F_REP / 2 - 1)) * (uint32_t)(2 * F_REPW * 4);
    static_assert((F_REP & 1) == 0 && (F_REPW & 1) == 1, "two arrays per replica, the second one an odd number of words behind the first");
#else
    lds_u8 *const wrep = (lds_u8 *)pwin + rep * (uint32_t)(F_REPW * 4);
#endif
    lds_u8 *const stage = (lds_u8 *)s_stage[wave] + F_PAD;          // the run starts here
    int32_t pw_base = 0;                                            // anchor of the wave's packed window
    int pw_tiles = F_FLUSH;                                         // tiles added since the last fold (forces an anchor for the first tile)
    // the wave's granule of the event list (see the insertion events below)
    const unsigned ev_shard = blockIdx.x & (EV_SHARDS - 1);
    amp_ins_event *const ev_list = eb.ev + (size_t)ev_shard * (size_t)eb.cap;
    unsigned long long ev_base = 0;
    uint32_t ev_left = 0;

    // folds the wave's packed window into the block's 32-bit window (or the global table) and clears it
    auto fold = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the asm adds of the counting phase are invisible to the compiler's wait counts
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
                    else if ((uint32_t)p < G) { atomicAdd(&counts[(size_t)p * AMP_NSYM + c], c4[c]); F_GLOB; }
                }
            }
        }
        wave_sync();
    };
    // the slots a wave reserved on the event list and did not use are marked (ref_pos = -1: dropped at read-out)
    auto pad_events = [&]() {
        if ((uint32_t)lane < ev_left && (long long)(ev_base + (unsigned)lane) < eb.cap) ev_list[ev_base + (unsigned)lane] = amp_ins_event{-1, 0u, 0, 0};
    };

    // ---- software pipeline: while tile t is computed from registers, the bytes of tile t + 1 are on their way and the
    // header of tile t + 2 is being read.  Loads of one tile form a chain header -> CIGAR words -> primer-table entries;
    // each link is issued one stage after the link before it has been waited for, so nothing in the loop waits for a
    // load it has just issued ------------------------------------------------------------------------------------------
    struct Hdr { int32_t pos, tlen; uint32_t lseq, flag, c0, c1, o8; };      // as loaded
    // ... and as kept between tiles: lf = l_seq (saturated at 0xFFFF: anything that long leaves the fast path anyway) |
    // paired << 16 | reverse << 17 | the template-length test of A:452 << 18 | number of CIGAR ops (saturated at 7) << 19
    struct HdrP { int32_t pos; uint32_t lf, c0, o8;
        __device__ uint32_t lseq() const { return lf & 0xFFFFu; }
        __device__ uint32_t nops() const { return (lf >> 19) & 7u; }
        __device__ uint32_t flag() const { return ((lf >> 16) & 1u) | (((lf >> 17) & 1u) << 4); }      // bits 0x1 and 0x10 of FLAG
        __device__ bool isize_flag() const { return (lf >> 18) & 1u; } };
    auto pack_hdr = [&](const Hdr &h) {
        const uint32_t n = h.c1 - h.c0, at = (uint32_t)(h.tlen < 0 ? -(int64_t)h.tlen : (int64_t)h.tlen);
        const bool isz = ((int64_t)at - P.max_primer_len) > (int64_t)h.lseq;                                  // A:452
        return HdrP{h.pos, (h.lseq > 0xFFFFu ? 0xFFFFu : h.lseq) | ((h.flag & 1u) << 16) | (((h.flag >> 4) & 1u) << 17) | ((isz ? 1u : 0u) << 18) | ((n > 7u ? 7u : n) << 19),
                    h.c0, h.o8};
    };
    struct Cg { uint32_t w[5]; };
    struct Geo { uint32_t np, phi, row, Tq; int ntake; bool solo, taken, fastq; };
    struct Tabs { int32_t L, R; };
    struct Bytes { uint4 raws[F_STAGE / 2048]; };
    struct Shape { Cig2 s; bool ok; int32_t refspan; };
    auto load_hdr = [&](int64_t t0) {
        Hdr h{0, 0, 0u, 0u, 0u, 0u, 0u};
        const int64_t i = t0 + lane;
        if (i < wend) {
            h.pos = rd.pos[i]; h.flag = rd.flag[i]; h.tlen = rd.tlen[i]; h.lseq = rd.lseq[i];
            h.c0 = rd.cig_off32[i]; h.c1 = rd.cig_off32[i + 1]; h.o8 = rd.seq_off8[i];
        }
        return h;
    };
    // the first five CIGAR words (lanes past the block's reads have none)
    auto load_cig = [&](const HdrP &h) {
        Cg c{{0u, 0u, 0u, 0u, 0u}};
        const uint32_t nops = h.nops();
        if (nops >= 1u && nops <= 5u) {
#pragma unroll
            for (uint32_t k = 0; k < 5u; ++k) c.w[k] = rd.cig[h.c0 + (k < nops ? k : 0u)];        // (words past the read's own repeat its first)
        }
        return c;
    };
    // the tile's run: the bytes of its leading reads, at most F_STAGE quality bytes (64 reads of up to 152 bases always
    // fit; the lanes behind a longer read do not, and go to the general pass like the long read itself)
    auto geometry = [&](const HdrP &h, int64_t t0, uint32_t &m0) {
        Geo g;
        const bool valid = t0 + lane < wend;
        const uint32_t hl = h.lseq();
        const bool shortq = valid && hl >= 1u && hl <= (uint32_t)F_MAXLEN;
        // pieces start phi bases before the read (its coordinates below are shifted by phi); np of them cover it
        g.phi = shortq ? phi_lane : 0u;
        g.np = shortq ? (hl + g.phi + 15u) >> 4 : 1u;
        m0 = __builtin_amdgcn_readfirstlane(h.o8);
        g.row = (h.o8 - m0) * 8u;                                            // byte offset of the read's qualities in the run
        const uint32_t nch = (hl + 7u) >> 3;
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
        return g;
    };
    // The shape of a read the fast path takes: one match op ("150M"), or two around ONE insertion / deletion
    // ("70M2I78M", "70M3D80M"), with or without soft clips at the ends ("12S138M", "5S70M2I61M12S"); refspan = its
    // reference length (A:451 looks the right table up at its last position)
    auto shape_of = [&](const HdrP &h, const Cg &c, bool fastq) {
        Shape r;
#if AMP_F4_BF
        bool ok;
        r.s = cig2_of(bf_from_words5((int)h.nops(), c.w[0], c.w[1], c.w[2], c.w[3], c.w[4], (int32_t)h.lseq(), F_MAXINS, F_MAXDEL, ok));
        r.ok = ok & fastq;
        if (!r.ok) r.s = Cig2{0u, 0, 0, 0, 0, 0, 0, false};
#else
        r.s = Cig2{0u, 0, 0, 0, 0, 0, 0, false};
        const int nops = (int)h.nops();
        r.ok = fastq && nops >= 1 && nops <= 5 && cig2_from_words5(nops, c.w, (int32_t)h.lseq(), r.s);
        if (r.ok && ((r.s.kind == 1 && r.s.k > F_MAXINS) || (r.s.kind == 2 && r.s.k > F_MAXDEL))) r.ok = false;
        if (!r.ok) r.s = Cig2{0u, 0, 0, 0, 0, 0, 0, false};
#endif
        r.refspan = r.ok ? r.s.m1 + r.s.m2 + (r.s.kind == 2 ? r.s.k : 0) : 1;               // (m2 = 0 without an indel; soft clips cover no reference)
        return r;
    };
    // the two primer-table entries of A:450-451
    auto load_tabs = [&](const HdrP &h, const Shape &sh) {
        Tabs t{-1, -1};
        const bool in_ref = (uint32_t)h.pos < G && (uint32_t)(h.pos + sh.refspan - 1) < G;
        if (sh.ok && P.do_trim && in_ref) { t.L = P.max_end[h.pos]; t.R = P.min_start[h.pos + sh.refspan - 1]; }
        return t;
    };
    // the tile's quality bytes by LDS-DMA (lane l moves bytes [1024 s + 16 l, + 16) of the run to the same offset of
    // the staging buffer) and its packed bases (16 bytes per lane and load)
    auto issue_bytes = [&](const Geo &g, uint32_t m0) {
        Bytes x;
        const uint8_t *qrun = rd.qual + (int64_t)m0 * 8;
        const uint8_t *srun = rd.seq + (int64_t)m0 * 4;
        // lanes past the run re-read its end
        const uint32_t lastq = g.Tq ? (g.Tq - 1u) & ~15u : 0u, lasts = g.Tq ? ((g.Tq >> 1) - 1u) & ~15u : 0u;
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 1024; ++sl) {
            uint32_t off = (uint32_t)(sl * 1024 + lane * 16);
            off = off < lastq ? off : lastq;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qrun + off),
                                             (__attribute__((address_space(3))) void *)(stage + sl * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 2048; ++sl) {
            uint32_t off = (uint32_t)(sl * 1024 + lane * 16);
            off = off < lasts ? off : lasts;
            const uint32_t *sp = (const uint32_t *)(srun + off);
            x.raws[sl] = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        }
        return x;
    };
    // Results of a tile are STORED ONE TILE LATER, right behind the wait at the top of the loop: stores and loads
    // retire through one in-order counter, so a store issued at the end of a tile would make that wait
    // last until the store has reached memory.
    // results of a tile, packed: meta = ncig | status << 8 | trim flags << 16 | P_* bits
    enum : uint32_t { P_STORED = 1u << 24, P_LIST = 1u << 25, P_STATUS_ONLY = 1u << 26 };
    struct Pend { uint32_t i, slot_lo; int32_t pos, reflen; uint32_t meta, cw0, cw1, cw2, cw3, cw4; };
    Pend pend{0u, 0u, 0, 0, 0u, 0u, 0u, 0u, 0u, 0u};
    auto store_pending = [&](const Pend &r) {
#if defined(AMP_DEV) && defined(AMP_ABL)
        if (AMP_ABL & 8) return;
#endif
        const uint32_t ncig = r.meta & 0xFFu;
        if (r.meta & P_STORED) {
            uint32_t *home = out.new_cig + ((size_t)r.slot_lo + 3 * (size_t)r.i);
            if (ncig > 0u) home[0] = r.cw0;
            if (ncig > 1u) home[1] = r.cw1;
            if (ncig > 2u) home[2] = r.cw2;
            if (ncig > 3u) home[3] = r.cw3;
            if (ncig > 4u) home[4] = r.cw4;
            if (out.new_pos) out.new_pos[r.i] = r.pos;
            if (out.new_ncig) out.new_ncig[r.i] = ncig;
            if (out.ref_len) out.ref_len[r.i] = r.reflen;
            if (out.trim_flags) out.trim_flags[r.i] = (uint8_t)(r.meta >> 16);
            if (out.status) out.status[r.i] = (uint8_t)(r.meta >> 8);
        }
        // hand-over to the general pass: the block's segment of the list (any order: the general kernel's window follows
        // the reads it is given, and a block's reads lie within a few hundred positions of each other)
        const bool has = (r.meta & P_LIST) != 0;
        const unsigned long long m = __ballot(has);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add((lds_u32 *)&s_gcur, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (has) glist[(size_t)rb + base + __popcll(m & ((1ull << lane) - 1ull))] = r.i | ((r.meta & P_STATUS_ONLY) ? GL_STATUS_ONLY : 0u);
        }
    };

    uint32_t pw_lim = 0;
    auto count_piece = [&](const uint4 &q, const uint2 &sq, int32_t j0, int32_t qa_, int32_t qb_, int32_t dbase_) -> bool {
        return fast_count_piece(q, sq, j0, qa_, qb_, dbase_, mqb, pw_lim, (uint32_t)(uintptr_t)wrep);
    };

#ifndef AMP_F_STAGGER
#define AMP_F_STAGGER 4000
#endif
    if (AMP_F_STAGGER > 0) {       // (the waves of a block start together and take equally long per tile: left alone they ask for memory and compute in step. Wave w starts w steps late: 0.243 -> 0.230 ms on the bench batch)
        const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)AMP_F_STAGGER * (unsigned)wave;
        while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(8);
    }
    // ---- prologue: header and bytes of the first tile, header of the second ----------------------------------------
    // Loop-carried state: hN/cN/gN/xN/m0N belong to the NEXT tile (its bytes are in flight), hN2 is the header of the one
    // after it.  They are renamed to "this tile" at the top of the loop, behind the wait, so that no register with a
    // load in flight is touched between the issue of a tile's loads and that wait (the compiler answers every such
    // touch with s_waitcnt vmcnt(0), which would wait for the bytes just requested).
    uint32_t tkN = take_ticket(), tkN2 = 0;
    int64_t i0 = rb, i1 = rb + 64 * (int64_t)tkN, i2 = rb;
    int tile_no = 0; (void)tile_no;
    HdrP hN = pack_hdr(load_hdr(i1));
    Hdr hN2{0, 0, 0u, 0u, 0u, 0u, 0u};
    Cg cN = load_cig(hN);
    uint32_t m0N = 0;
    Geo gN = geometry(hN, i1, m0N);
    Bytes xN = issue_bytes(gN, m0N);
    tkN2 = take_ticket();
    i2 = rb + 64 * (int64_t)tkN2;
    hN2 = load_hdr(i2);
    while (tkN < n_tb) {
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the DMA of this tile's qualities has landed, its other loads too
        F_STAMP(1);
#if defined(AMP_DEV) && defined(AMP_ABL)
        if (AMP_ABL & 32) { if (((tile_no++) + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
        const HdrP h = hN;
        const Cg cA = cN;
        const Geo g = gN;
        const Bytes xA = xN;
        hN = pack_hdr(hN2);
        i0 = i1; i1 = i2; tkN = tkN2;
        const int64_t i = i0 + lane;
        const int32_t pos = h.pos;
        const uint32_t lseq = h.lseq(), flag = h.flag(), c0 = h.c0, o8 = h.o8;
        const uint32_t np = g.np, phi = g.phi;
        const bool solo = g.solo, taken = g.taken, fastq = g.fastq;
        // the primer-table entries of this tile (they hang on its CIGAR words): a short wait behind the row reads below
        const Shape shp = shape_of(h, cA, fastq);
        const Tabs tA = load_tabs(h, shp);
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
        pw_lim = (int64_t)G - pw_base >= (int64_t)F_PW ? (uint32_t)F_PW : (uint32_t)(G > (uint32_t)pw_base ? G - (uint32_t)pw_base : 0u);
        // ---- rows: slot k of the lane holds piece (k + rot) mod np of its read (slots >= np: a copy of the last
        // piece and an index past the read, which every range test below excludes) -------------------------------
        const uint32_t rot = (uint32_t)lane % np;
        const uint8_t *qrow = rd.qual + (int64_t)o8 * 8;
        const uint8_t *srow = rd.seq + (int64_t)o8 * 4;
        const int32_t lrow = fastq ? (int32_t)g.row - (int32_t)phi : 0;        // >= -8: the pad in front of the run
        const lds_u8 *const lq = stage + (fastq ? (int32_t)g.row : 0);         // the read's qualities in the staging buffer
        uint4 q16[F_NP];
        uint2 s8[F_NP];
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
        // ---- primer clips in closed form (A:450-558) ---------------------------------------------------------------
        Cig2 s = shp.s;
        const bool shaped = shp.ok;
        const bool in_ref = (uint32_t)pos < G && (uint32_t)(pos + shp.refspan - 1) < G;          // A:450-451
        const bool rev = (flag & 0x10u) != 0;
        // query index of the first inserted base / of the base behind the deletion, and of the second match segment:
        // no clip moves them while the indel survives
        const int32_t q_ins = s.kind ? s.a + s.m1 : 0, q_seg2 = q_ins + (s.kind == 1 ? s.k : 0);
        TrimState ts{pos, 1, 0u, 0};
#if AMP_F4_BF
        {
            const bool trim = shaped & (P.do_trim != 0), use = trim & in_ref;
            ts.err = (trim & !in_ref) ? AMP_RS_INDEX_REF : 0;
            int32_t p2 = pos; uint32_t f2 = 0u;
            const Bf sp = bf_trim_primers(bf_of(s), p2, f2, flag, h.isize_flag(), (int32_t)lseq, tA.L, tA.R);
            s = cig2_of(bf_pick(use, sp, bf_of(s))); ts.pos = use ? p2 : pos; ts.flags = use ? f2 : 0u;
        }
        const bool scan = shaped & (P.do_trim != 0) & (ts.err == 0) & !s.punt;
        // aligned-quality window [lo, hi) in PIECE coordinates (query index + phi)
        int32_t lo, qlen;
        bf_quality_window(bf_of(s), (int32_t)lseq, lo, qlen);
        lo = scan ? lo + (int32_t)phi : 0; qlen = scan ? qlen : 0;
#else
        if (shaped && P.do_trim) {
            if (!in_ref) ts.err = AMP_RS_INDEX_REF;
            else cig2_trim_primers_isize(ts, flag, h.isize_flag(), (int32_t)lseq, s, tA.L, tA.R);
        }
        const bool scan = shaped && P.do_trim && !ts.err && !s.punt;
        // aligned-quality window [lo, hi) in PIECE coordinates (query index + phi)
        int32_t lo = 0, qlen = 0;
        if (scan) { cig2_quality_window(s, (int32_t)lseq, lo, qlen); lo += (int32_t)phi; }
#endif
        const int32_t hi = lo + qlen;
        // Bytes the slots cannot give (they are rotated per lane), read from the staged qualities while they are still there:
        // the 3' end's shrinking windows (A:575-576, A:637-638) need at most W-1 bytes; reads with an indel need the
        // qualities around the insertion and GROUP B = the 16 bases from the 8-aligned start of the second segment, for
        // the piece that holds bases of both segments (its second part is counted from this copy)
        const int32_t first = !scan ? 0 : ((rev || qlen < W) ? lo : lo + qlen - W + 1) - (int32_t)phi;      // query index
        const int32_t tab = first & ~7, g_ins = q_ins & ~7, g_b = q_seg2 & ~7;
        const amp_u32x2 tw0 = *(const lds_u32x2 *)(lq + tab), tw1 = *(const lds_u32x2 *)(lq + tab + 8);
        const amp_u32x2 iq0 = *(const lds_u32x2 *)(lq + g_ins), iq1 = *(const lds_u32x2 *)(lq + g_ins + 8);
        const amp_u32x2 bq0 = *(const lds_u32x2 *)(lq + g_b), bq1 = *(const lds_u32x2 *)(lq + g_b + 8);
        wave_sync();
#pragma unroll
        for (int sl = 0; sl < F_STAGE / 2048; ++sl) *(lds_u32x4 *)(stage + sl * 1024 + lane * 16) = amp_u32x4{xA.raws[sl].x, xA.raws[sl].y, xA.raws[sl].z, xA.raws[sl].w};
        wave_sync();
#if AMP_F4_LEAN
        {
            // the test for codes outside A C G T looks at whole pieces (nibbles_bad): pad nibbles of the staged rows (a row is
            // padded to 8 bases) and the 16 bytes behind the run become a valid code
            if (lane == 0) *(lds_u32x4 *)(stage + ((g.Tq >> 1) & ~3u)) = amp_u32x4{0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u};
            wave_sync();
            const uint32_t e = fastq ? lseq & 7u : 0u;                        // bases of the row's last group of 8 (0: the group is full)
            if (e) {
                lds_u32 *w = (lds_u32 *)(stage + (g.row >> 1) + 4u * (lseq >> 3));
                const uint32_t x = *w, xs = ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);      // nibble i at bit 4 i
                const uint32_t keepn = (1u << (4u * e)) - 1u;
                const uint32_t ys = (xs & keepn) | (0x11111111u & ~keepn);
                *w = ((ys & 0x0F0F0F0Fu) << 4) | ((ys >> 4) & 0x0F0F0F0Fu);
            }
            wave_sync();
        }
#endif
#pragma unroll
        for (int k = 0; k < F_NP; ++k) {
            uint32_t p = (uint32_t)k + rot;
            p = p >= np ? p - np : p;
            p = (uint32_t)k < np ? p : np - 1u;
            const lds_u8 *src = stage + (lrow >> 1) + (int32_t)(p * 8u);
            s8[k] = make_uint2(*(const lds_u32 *)src, *(const lds_u32 *)(src + 4));
        }
        const lds_u8 *const lsq = stage + (fastq ? (int32_t)(g.row >> 1) : 0) + (g_b >> 1);
        const uint2 bsq = make_uint2(*(const lds_u32 *)lsq, *(const lds_u32 *)(lsq + 4));
        wave_sync();                                 // every lane has its rows: the staging buffer may be overwritten
        store_pending(pend);                         // the previous tile's results
        // ---- next tile: its CIGAR words and bytes start moving now (its header arrived with this tile's bytes), and the
        // header of the tile behind it.  Nothing below touches them before the wait at the top of the loop.  No branch
        // around the loads of the bytes (behind the wave's last tile they fetch the first bytes of the batch) ---------
        tkN2 = take_ticket();                       // (an LDS access the compiler can see: before the DMA is in flight)
        i2 = rb + 64 * (int64_t)tkN2;
        cN = load_cig(hN);
        gN = geometry(hN, i1, m0N);
        xN = issue_bytes(gN, m0N);
        hN2 = load_hdr(i2);
        F_STAMP(2);          // staged, rows in registers, primer clips, next tile issued

        // ---- sliding-window scan: first failing window start (forward) / last failing window end (reverse) --
        int32_t ffmin = 0x7FFFFFFF, lemax = -1;
#if defined(AMP_DEV) && defined(AMP_ABL)
        if (P.do_trim && !(AMP_ABL & 4)) {
#else
        if (P.do_trim) {
#endif
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
        bool general = i < re && !shaped;               // (a lane whose bytes did not fit the run included)
        bool stored = false;
        uint32_t ncig = 0, cw[5] = {0u, 0u, 0u, 0u, 0u};
        int32_t reflen = 0;
        if (shaped) {
            // the read's first quality byte (0xFF = QUAL '*') is byte phi of piece 0, which sits in slot (np - rot) mod np
            uint32_t fb = phi ? q16[0].z : q16[0].x;
#pragma unroll
            for (int k = 1; k < F_NP; ++k) fb = ((uint32_t)k + rot == np) ? (phi ? q16[k].z : q16[k].x) : fb;
            if ((fb & 0xFFu) == 0xFFu || s.punt) {
                general = true;                   // QUAL '*': the generic code reports it (A:561-562, A:718); a shape the closed forms leave
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
#if AMP_F4_BF
                    { uint32_t f2 = ts.flags; s = cig2_of(bf_trim_quality(bf_of(s), ts.pos, f2, rev, iq, qlen)); ts.flags = f2; }
#else
                    cig2_trim_quality(ts, rev, iq, qlen, s);
#endif
                }
                if (s.punt) {
                    general = true;
                } else {
                    if (!ts.err) {
                        // [S a][op m1][I|D k][op m2][S c], absent parts left out: slot `ncig` takes the next part
                        const uint32_t part[5] = {((uint32_t)s.a << 4) | OP_S, ((uint32_t)s.m1 << 4) | s.op,
                                                  ((uint32_t)s.k << 4) | (s.kind == 1 ? OP_I : OP_D), ((uint32_t)s.m2 << 4) | s.op,
                                                  ((uint32_t)s.c << 4) | OP_S};
                        const bool has[5] = {s.a > 0, s.m1 > 0, s.kind != 0, s.kind != 0 && s.m2 > 0, s.c > 0};
#pragma unroll
                        for (int t = 0; t < 5; ++t) {
                            if (has[t]) {
#pragma unroll
                                for (int j = 0; j < 5; ++j) cw[j] = ncig == (uint32_t)j ? part[t] : cw[j];
                                ++ncig;
                            }
                        }
                        reflen = s.ref_len();
                    }
                    stored = true;
                    if (ts.err) ++n_err;
                }
            }
        }
        const bool counted = stored && !ts.err && P.do_count;
        F_STAMP(4);          // quality clip, results
        // ---- counting (A:709-753): the counted query ranges [qa1, qb1) and [qa2, qb2) in piece coordinates, the window
        // offset of piece coordinate 0 for each of them ------------------------------------------------------------------
        const bool two = counted && s.kind != 0;
        const int32_t qa1 = counted ? s.a + (int32_t)phi : 0, qb1 = counted ? qa1 + s.m1 : 0;
        const int32_t qa2 = two ? qb1 + (s.kind == 1 ? s.k : 0) : qb1, qb2 = two ? qa2 + s.m2 : qa2;
        const int32_t pos2 = ts.pos + s.m1 + (s.kind == 2 ? s.k : 0);              // reference position of the second segment
        bool bad_extra = false;
        // deletion: '-' at each of its positions (A:714-715), through the block's window
        if (two && s.kind == 2) {
            for (int32_t j = 0; j < s.k; ++j) {
                const int32_t r = ts.pos + s.m1 + j;
                const uint32_t d = (uint32_t)(r - bw_base);
                if ((uint32_t)r >= G) bad_extra = true;
                else if (d < (uint32_t)F_BW) lds_add_nt(bwin + 4 * F_BW + d, 1u);
                else atomicAdd(&counts[(size_t)r * AMP_NSYM + 5], 1u);
            }
        }
        // insertion (A:730-748): one event per maximal run of good-quality inserted bases (cig2_indels in amp_read.hpp;
        // the low base that ends a run is consumed, which changes nothing: it would be skipped anyway)
        {
            uint32_t good = 0;
            if (two && s.kind == 1) {
                const uint32_t o0 = ok_bits4(iq0.x, mqb) >> 7, o1 = ok_bits4(iq0.y, mqb) >> 7, o2 = ok_bits4(iq1.x, mqb) >> 7, o3 = ok_bits4(iq1.y, mqb) >> 7;
                // byte flags -> bits: bit b of a dword's nibble = bit 8 b of the flags
                auto nib = [](uint32_t o) { return ((o * 0x00204081u) >> 21) & 0xFu; };
                const uint32_t m16 = nib(o0) | (nib(o1) << 4) | (nib(o2) << 8) | (nib(o3) << 12);
                good = (m16 >> (uint32_t)(s.a + s.m1 - g_ins)) & ((1u << s.k) - 1u);       // (a clip may have taken the first inserted bases)
            }
            uint32_t runs = good & ~(good << 1);                              // first base of every run
            const unsigned long long em = __ballot(runs != 0u);
            if (em) {
                // Event slots come from the wave's GRANULE of the list: one returning atomic on the shard's cursor
                // reserves F_EVGRAN slots (a reservation per tile would serialise the chip on a few addresses)
                const uint32_t total = (uint32_t)__popcll(em);
                if (total > ev_left) {
                    pad_events();
                    unsigned long long nb = 0;
                    if (lane == 0) nb = atomicAdd(&ctr[16 + ev_shard], (unsigned long long)F_EVGRAN);
                    ev_base = __shfl(nb, 0); ev_left = F_EVGRAN;
                }
                if (runs) {
                    const int32_t q0 = s.a + s.m1, r2 = ts.pos + s.m1, ref_end = ts.pos + s.m1 + s.m2;
                    // the lane's first run takes its slot of the granule; further runs of one insertion (rare) go the slow way
                    const unsigned long long slot = ev_base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
                    const uint32_t rid = (uint32_t)(read_base + (uint64_t)i);
                    bool firstrun = true;
                    while (runs) {
                        const int32_t js = __builtin_ctz(runs);
                        runs &= runs - 1u;
                        const int32_t je = js + __builtin_ctz(~(good >> js));
                        int32_t elo, ehi;
                        if (je == s.k && s.m2 > 0 && r2 == 0) py_slice(q0 + js, q0 + je + 1, (int32_t)lseq, elo, ehi);   // A:735-736
                        else py_slice(q0 + js - 1, q0 + je, (int32_t)lseq, elo, ehi);              // A:738
                        int32_t ins_pos = je == s.k ? r2 : ref_end;                                // A:742 / A:739-740
                        ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;                               // A:744
                        const bool inside = (uint32_t)ins_pos < G;
                        if (!inside) bad_extra = true;
                        if (firstrun) {
                            if ((long long)slot < eb.cap) ev_list[slot] = inside ? amp_ins_event{ins_pos, rid, elo, ehi} : amp_ins_event{-1, 0u, 0, 0};
                            if (inside) {
                                const uint32_t d = (uint32_t)(ins_pos - bw_base);
                                if (d < (uint32_t)F_BW) lds_add_nt(bwin + 5 * F_BW + d, 1u);
                                else atomicAdd(&eb.ins_at[ins_pos], 1u);
                            }
                        } else if (inside) {
                            eb.record(ins_pos, rid, elo, ehi);
                        }
                        firstrun = false;
                    }
                }
                ev_base += total; ev_left -= total;
            }
        }
        uint32_t redo = 0;                        // pieces (slots) the careful loop has to do; bit F_NP = group B
#if AMP_F4_LEAN
        uint32_t okm[F_NP];                       // 16 good-quality bits per piece: the qualities are done with
#pragma unroll
        for (int k = 0; k < F_NP; ++k) okm[k] = ok_bits16(q16[k], mqb);
        const uint32_t okB = ok_bits16(make_uint4(bq0.x, bq0.y, bq1.x, bq1.y), mqb);
#endif
        // group B: the part of the second segment that shares a piece with the first
        int32_t xbe = 0, jb = 0;
        bool has_b = false;
        if (two) {
            const int32_t jstar = (qb1 - 1) & ~15;                   // the piece that holds the first segment's last base
            has_b = jstar + 16 > qa2 && qb2 > qa2;
            xbe = qb2 < jstar + 16 ? qb2 : jstar + 16;
            jb = g_b + (int32_t)phi;
        }
        // The bases go into the wave's packed window, F_PW positions from pw_base.  Reads of one tile usually lie within a
        // few positions of each other; where the batch steps from one pile of reads to the next they do not, and the
        // tile is counted in PASSES: every pass takes the lanes whose counted positions all lie inside the window, then the
        // window is folded and anchored again at the first lane that is left.  (Without this the lanes behind the step went
        // through the careful loop, base by base from global memory: one such tile cost as much as a dozen others.)
        const int32_t end_pos = two ? pos2 + s.m2 : ts.pos + s.m1;                 // one past the last counted position
        bool todo = counted;
#if defined(AMP_DEV) && defined(AMP_ABL)
        if (AMP_ABL & 2) todo = false;
#endif
        for (bool first_pass = true;; first_pass = false) {
            const unsigned long long tm = __ballot(todo);
            if (!tm) break;
            const int lead = __builtin_ctzll(tm);                                   // sorted input: the smallest position that is left
            if (!first_pass) {
                fold();
                const int32_t lead_pos = __builtin_amdgcn_readlane(pos, lead);
                pw_base = (lead_pos < 16 ? 0 : lead_pos - 16) & ~15; pw_tiles = 1;
                pw_lim = (int64_t)G - pw_base >= (int64_t)F_PW ? (uint32_t)F_PW : (uint32_t)(G > (uint32_t)pw_base ? G - (uint32_t)pw_base : 0u);
            }
            // pieces are 16 positions wide: 16 positions of slack at both ends.  The lead lane always goes (a piece of
            // its read that does not fit -- at the very end of the reference -- is left to the careful loop)
            const bool fits = ts.pos - pw_base >= 16 && end_pos - pw_base + 16 <= (int32_t)pw_lim;
            const bool now = todo && (fits || lane == lead);
            const int32_t a1 = now ? qa1 : 0, b1 = now ? qb1 : 0, a2 = now ? qa2 : 0, b2 = now ? qb2 : 0;
            const int32_t dbase1 = ts.pos - pw_base - qa1, dbase2 = pos2 - pw_base - qa2;
#if AMP_F4_LEAN
            const int32_t lim16 = (int32_t)pw_lim - 16;
            if (__ballot(has_b && now)) {
                const uint32_t mB = (has_b & now) ? okB & range_bits16(qa2 - jb, xbe - jb) : 0u;
                redo |= F_COUNT5(bsq, mB, dbase2 + jb, lim16, (uint32_t)(uintptr_t)wrep) << F_NP;
            }
#else
            if (__ballot(has_b && now)) {
                if (has_b && now && count_piece(make_uint4(bq0.x, bq0.y, bq1.x, bq1.y), bsq, jb, qa2, xbe, dbase2)) redo |= 1u << F_NP;
            }
#endif
#pragma unroll
            for (int k = 0; k < F_NP; ++k) {
                uint32_t p = (uint32_t)k + rot;
                p = p >= np ? p - np : p;
                p = (uint32_t)k < np ? p : np;
                const int32_t j0 = (int32_t)(p * 16u);
                const bool second = j0 >= qb1;                              // a piece behind the first segment belongs to the second
                // (the empty asm keeps the compiler from computing the pass-independent half of every piece in front of
                // the pass loop, which costs eighty registers this kernel does not have)
#if AMP_F4_LEAN
                redo |= F_COUNT5(s8[k], okm[k] & range_bits16((second ? a2 : a1) - j0, (second ? b2 : b1) - j0), (second ? dbase2 : dbase1) + j0, lim16,
                                 (uint32_t)(uintptr_t)wrep) << k;
#else
                asm volatile("" : "+v"(q16[k].x), "+v"(q16[k].y), "+v"(q16[k].z), "+v"(q16[k].w), "+v"(s8[k].x), "+v"(s8[k].y));
                if (count_piece(q16[k], s8[k], j0, second ? a2 : a1, second ? b2 : b1, second ? dbase2 : dbase1)) redo |= 1u << k;
                __builtin_amdgcn_sched_barrier(0);          // one piece at a time: interleaving them costs registers the kernel does not have
#endif
            }
            todo = todo && !now;
        }
        F_STAMP(5);          // counting
        bool want_status = bad_extra;
        if (__ballot(redo != 0u)) {
            // careful loop (rare): bases of the flagged pieces one by one, straight from memory into the 32-bit counters
            if (redo) {
                const int32_t a1 = qa1 - (int32_t)phi, b1 = qb1 - (int32_t)phi, a2 = qa2 - (int32_t)phi, b2 = qb2 - (int32_t)phi;   // query indices
                auto careful = [&](int32_t x0, int32_t x1, int32_t qa_q, int32_t rp0) {
                    for (int32_t q = x0; q < x1; ++q) {
                        if ((int32_t)qrow[q] < mq) continue;
                        const uint32_t sb = srow[q >> 1];
                        const uint32_t col = col_of_code((q & 1) ? (sb & 15u) : (sb >> 4));
                        const int32_t rp = rp0 + (q - qa_q);
                        const uint32_t d = (uint32_t)(rp - bw_base);
                        if (col > 4u || (uint32_t)rp >= G) want_status = true;
                        else if (d < (uint32_t)F_BW && col < (uint32_t)F_NPL) lds_add_nt(bwin + col * F_BW + d, 1u);
                        else atomicAdd(&counts[(size_t)rp * AMP_NSYM + col], 1u);
                    }
                };
                for (int k = 0; k < F_NP; ++k) {
                    if (!((redo >> k) & 1u)) continue;
                    uint32_t p = (uint32_t)k + rot;
                    p = p >= np ? p - np : p;
                    const int32_t j0 = (int32_t)(p * 16u) - (int32_t)phi;
                    const bool second = j0 + (int32_t)phi >= qb1;
                    const int32_t sa = second ? a2 : a1, sb_ = second ? b2 : b1;
                    careful(j0 < sa ? sa : j0, j0 + 16 < sb_ ? j0 + 16 : sb_, sa, second ? pos2 : ts.pos);
                }
                if ((redo >> F_NP) & 1u) careful(a2, xbe - (int32_t)phi, a2, pos2);
            }
        }

        // ---- results and hand-over to the general pass: kept for the next turn of the loop ---------------------------
        {
            uint32_t meta = ncig | ((uint32_t)ts.err << 8) | ((ts.err ? 0u : ts.flags) << 16) | (stored ? P_STORED : 0u);
            if (general) meta |= P_LIST;
            else if (counted && want_status) meta |= P_LIST | P_STATUS_ONLY;      // a base could not be counted: exact status wanted
            pend = Pend{(uint32_t)i, c0, ts.pos, reflen, meta, cw[0], cw[1], cw[2], cw[3], cw[4]};
        }
        F_STAMP(6);          // careful loop
    }
    F_EPIW(0);
    store_pending(pend);
    pad_events();
    F_EPI(1);
    if (pw_tiles && n_tb) fold();
    F_EPI(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (asm adds into the block's window: deletions, insertion tally, careful loop)
    __syncthreads();
#if defined(AMP_DEV) && defined(AMP_ABL)
    if (AMP_ABL & 16) return;
#endif
    for (int i = tid; i < F_BPL * F_BW; i += F_WAVES * 64) {
        const uint32_t v = bwin[i];
        if (v) {
            const int pl = i / F_BW, d = i - pl * F_BW;
            const uint32_t p = (uint32_t)(bw_base + d);
            if (p < G) {
                if (pl < F_NPL) atomicAdd(&counts[(size_t)p * AMP_NSYM + pl], v);
                else if (pl == 4) atomicAdd(&counts[(size_t)p * AMP_NSYM + 5], v);      // '-'
                else atomicAdd(&eb.ins_at[p], v);
            }
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
    F_STAMP_OUT;
    if (tid == 0) { gcnt[blockIdx.x] = s_gcur; if (s_gcur) eb.ctr[29] = (unsigned long long)P.epoch; }      // (every block writes the same value)
}

// Dense list of the reads the fast kernel handed over + the geometry of the general pass.  Block b places the
// segment of fast-kernel block b behind the totals of the blocks before it (the per-block counts are a KB in L2).
__global__ void __launch_bounds__(256)
k_gcompact(const uint32_t *__restrict__ glist, const uint32_t *__restrict__ gcnt, int reads_per_block, int64_t n_reads,
           uint32_t *__restrict__ dense, GenGeo *geo, uint32_t gen_grid, unsigned long long *ctr,
           const uint32_t *__restrict__ cig_off32, uint32_t *__restrict__ llist, uint32_t *__restrict__ lpos, int long_max_ops) {
    __shared__ uint32_t s_part[4], s_lw[4], s_lbase;
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) acc += gcnt[b];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    const uint32_t off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const uint32_t cnt = gcnt[blockIdx.x];
    const int64_t sbeg = (int64_t)blockIdx.x * reads_per_block;
    const uint32_t *src = glist + (sbeg < n_reads ? sbeg : 0);
    for (uint32_t k = tid; k < cnt; k += 256) dense[off + k] = src[k];
    if (long_max_ops) {
        // the reads with more ops than the tile kernel's columns hold (it would only pass them on) and no more than k_long's rows:
        // flagged in the dense list and listed, in order, for k_long (one reservation per block)
        // 1 = more ops than the tile kernel's columns hold, 2 = any read k_long can take.  A segment that is mostly made of
        // the first kind (a Nanopore-like batch) hands over every read it can: the few others would leave the tile kernel
        // one or two lanes of work per 64-read tile (79 us on 200,000 such reads)
        const auto kind = [&](uint32_t k) -> uint32_t {
            if (k >= cnt) return 0u;
            const uint32_t e = src[k];
            if (e & GL_STATUS_ONLY) return 0u;
            const uint32_t i = e & GL_INDEX_MASK, nops = cig_off32[i + 1] - cig_off32[i];
            if ((int)nops > long_max_ops) return 0u;
            return (int)nops + 3 > T_MAXOPS ? 1u : 2u;
        };
        uint32_t strict = 0, elig = 0;
        for (uint32_t k = tid; k < cnt; k += 256) { const uint32_t kd = kind(k); strict += kd == 1u ? 1u : 0u; elig += kd ? 1u : 0u; }
        for (int o = 32; o > 0; o >>= 1) { strict += __shfl_down(strict, o); elig += __shfl_down(elig, o); }
        __syncthreads();
        if ((tid & 63) == 0) { s_lw[tid >> 6] = strict; s_part[tid >> 6] = elig; }      // (s_part: the offsets it held are in `off`)
        __syncthreads();
        const uint32_t n_strict = s_lw[0] + s_lw[1] + s_lw[2] + s_lw[3], n_elig = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        const bool promote = (unsigned long long)n_strict * 4ull >= (unsigned long long)cnt * 3ull;
        const uint32_t total = promote ? n_elig : n_strict;
        const auto is_long = [&](uint32_t k) -> bool { const uint32_t kd = kind(k); return kd == 1u || (promote && kd == 2u); };
        if (tid == 0 && cnt > total) atomicAdd(&ctr[28], (unsigned long long)(cnt - total));      // entries left to the tile kernel
        if (total) {                                   // (uniform over the block)
            if (tid == 0) s_lbase = (uint32_t)atomicAdd(&ctr[26], (unsigned long long)total);
            uint32_t run = 0;
            for (uint32_t k0 = 0; k0 < cnt; k0 += 256) {
                const uint32_t k = k0 + tid;
                const bool lg = is_long(k);
                const unsigned long long m = __ballot(lg);
                __syncthreads();
                if ((tid & 63) == 0) s_lw[tid >> 6] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t before = run;
                for (int w = 0; w < (tid >> 6); ++w) before += s_lw[w];
                if (lg) {
                    const uint32_t at = s_lbase + before + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull));
                    llist[at] = src[k] & GL_INDEX_MASK; lpos[at] = off + k;
                    dense[off + k] = src[k] | GL_LONG;
                }
                run += s_lw[0] + s_lw[1] + s_lw[2] + s_lw[3];
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const uint32_t n_list = off + cnt;
        const uint32_t tiles = (n_list + TILE - 1) / TILE;
        uint32_t tpb = (tiles + gen_grid - 1) / gen_grid;
        tpb = ((tpb + T_WAVES - 1) / T_WAVES) * T_WAVES;
        if (tpb < (uint32_t)T_WAVES) tpb = T_WAVES;
        geo->n_list = n_list; geo->tpb = tpb; geo->n_seg = (tiles + tpb - 1) / tpb; geo->live_counted = long_max_ops ? 1u : 0u;
        ctr[7] = n_list;                          // (amp_debug_counters: reads of the last batch that took the general pass)
    }
}

static inline int fast_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out, uint32_t *counts,
                              const EventBuf &eb, uint32_t *glist, uint32_t *gcnt, const FastGrid &fg, hipStream_t stream, uint32_t *dbg) {
    const unsigned g = (unsigned)fg.grid, t = F_WAVES * 64;
    const int rpb = (int)fg.rpb;
    switch (P.window) {
        case 1: k_fast<1><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        case 2: k_fast<2><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        case 3: k_fast<3><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        case 4: k_fast<4><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        case 5: k_fast<5><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        case 6: k_fast<6><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        case 7: k_fast<7><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
        default: k_fast<8><<<g, t, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb) F_DBG_ARG(dbg)); break;
    }
    return (int)hipGetLastError();
}

}  // namespace amp
