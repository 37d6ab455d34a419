// amp_fast6.hpp -- the fast kernel, third generation (variant 6): trim + pileup of the reads whose CIGAR has the shape
// [S a][M m1]([I|D k][M m2])[S c] and at most 160 bases, one lane per read, every byte of the batch loaded once.  CDNA4 / gfx950.
//
// Written to test two readings of what bounds the first two generations: "a tile pays the trims of a read
// with an indel for 64 reads of which six have one" and "fewer, cheaper instructions make the pass faster".  Outcome: 23 % fewer
// vector instructions than amp_fast.hpp and the same time -- row traffic, the vector pipe and the LDS round trips each need a
// third of the kernel's time and overlap only partly at two waves per SIMD.  The kernel is therefore an opt-in variant
// (amp_set_kernel_variant(6)); it gives the same results as the others (tests/test_gpu_parity.py runs it through every test).
// What tools/micro/valu_rate3.hip measures: not every instruction costs the same -- add / sub / shift / and / or / compare /
// select / v_bitop3 issue in ~2.5 cycles with two waves on the SIMD, the three-operand and byte instructions (v_and_or,
// v_perm, v_dot4, SDWA, v_bfm, v_med3, v_bcnt, v_mul_u32_u24, v_readlane) in ~4.5, v_qsad_pk_u16_u8 in 17.  A tile of 64 reads
// cost ~8,500 such cycles, half of them in logic that only a read with an indel needs.  This generation spends fewer:
//   * CLASSES.  A block first sorts its reads by the number of CIGAR ops into two lists: "simple" (one or two ops: a
//     match op with at most one soft clip) and "indel" (three to five ops); anything else goes straight to the general pass.
//     A tile is 64 reads of ONE class, so nine tiles in ten of an amplicon run execute the closed-form trims of a read
//     without indel (a few dozen instructions instead of ~300) and none of the deletion / insertion-event code.
//   * ROWS BY GATHER.  The reads of a tile come from a list, so their bytes are moved row by row: LDS-DMA with one
//     global address per lane and immediate offsets for the 16-byte chunks of a row (no address arithmetic per instruction;
//     two lanes per row and instruction for the qualities = 32 contiguous bytes, tools/micro/dma_patterns.hip: 4.7 TB/s
//     against 5.4 for one contiguous run and 3.9 with one lane per row).  The LDS image is chunk-major, so a lane reads
//     16 bytes of its row with one conflict-free ds_read_b128 at an immediate offset.
//   * THREE PASSES over a tile's bytes, all from LDS:
//       1a  qualities -> failing-window bits (v_qsad_pk_u16_u8, sign bits gathered by v_perm + v_dot4), in the natural
//           order of the pieces: the 160 bits of a read sit in five registers and the first / last failing window is
//           found once per read, not piece by piece
//       1b  qualities + packed bases -> the base codes of the low-quality bases are zeroed in LDS (byte flags
//           "quality >= min_quality" turned into nibble masks with two byte permutes and four shifts per eight bases:
//           no bit gathering, no bit -> byte expansion); the two pieces that hold the ends of the counted range are
//           masked once more when the clips are known
//       2   masked codes -> counters.  The shift count of a base's counter byte comes from ONE byte permute per four
//           bases whose selector is the code itself: A C G -> 8 16 24, T (code 8) -> 0 through the permute's sign
//           selector, a zeroed code -> 31, any other code -> 0xFF (bit 7 marks the read for the careful loop, which adds
//           its N calls).  The value shifted is the constant 1: a masked base adds bit 31, which nothing reads.  So a
//           piece costs 16 SDWA shifts + 16 ds_add_u32 and ~25 other instructions, for every kind of read.
//     The quality buffer is free after 1b: the next tile's quality rows are requested then, and its packed bases are loaded by
//     the lanes into registers (the LDS only ever holds the MASKED bases of the tile being counted).
//   * The wave's packed window (one 32-bit word per reference position: T A C G counters of 8 8 8 7 bits + the bit the
//     masked bases hit, 4 replicas) is folded into the block's 32-bit window every 7 tiles at the latest.
// Results (new CIGAR, position, flags, counts, insertion events) are bit-identical to the other variants: same closed
// forms (amp_bf.hpp, fuzzed against the generic code on the CPU), same hand-over list for the general pass.
#pragma once

#include <type_traits>

#include "amp_fast5.hpp"

namespace amp {

// ablation builds (development: parts of the kernel switched off to time the rest; results are wrong on purpose)
#ifndef AMP_F6_ABL
#define AMP_F6_ABL 0
#endif
constexpr int F6_WAVES = 8;
constexpr int F6_NP = 10;                  // 16-base pieces of a row
constexpr int F6_MAXLEN = 16 * F6_NP;      // longest read taken
constexpr int F6_QB = 1024 * F6_NP;        // bytes of a wave's quality image (10 instructions x 64 lanes x 16 bytes)
constexpr int F6_SB = 512 * F6_NP;         // ... of its base image (5 instructions)
constexpr int F6_PW = 224;                 // reference positions covered by a wave's packed window
constexpr int F6_REP = 4;                  // replicas of it
constexpr int F6_REPW = F6_PW + 3;         // words per replica: replica r is skewed by 3 r banks (tools/micro/lds_pile.hip: with replica (lane >> 1) & 3
                                           // an add costs 4.0 - 4.7 LDS cycles on piles whose reads start within 3 .. 64 positions, against 8 for a skew of one bank)
constexpr int F6_BW = 480;                 // positions of the block's 32-bit window
constexpr int F6_FLUSH = 7;                // tiles between two folds: a counter gets at most 64 / F6_REP = 16 increments per tile, G has 7 bits
constexpr uint32_t F6_LUT_LO = 0xFF10081Fu, F6_LUT_HI = 0xFFFFFF18u;      // shift count by code: 0 -> 31, A -> 8, C -> 16, G -> 24; T (8) -> 0 by the sign selector

// LDS-DMA of 16 bytes per lane: global address g + OFF, LDS address m0 + OFF + 16 * lane (issued where the compiler cannot see it)
template <int OFF>
__device__ __forceinline__ void dma16_off(const void *g, uint32_t m0v_) {
    const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)m0v_);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(g), "s"(m0v), "n"(OFF) : "memory", "m0");
}
// the ten instructions of a tile's quality rows in one statement: 16 bytes per lane from (rows 32 .. 63 ? go : ge) + 32 j to
// lds + 1024 (2 j + (rows 32 .. 63)) + 16 lane (m0 + the instruction's offset + 16 lane).  The five instructions of a half of the
// rows follow each other: a 128-byte line of a row is touched by up to four of them, and the fewer lines a wave keeps alive in
// the 32 KB vector cache of its CU (eight waves stream through it), the fewer are fetched from L2 twice
__device__ __forceinline__ void dma_q_rows(const void *ge, const void *go, uint32_t lds_) {
    const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\ts_add_u32 m0, m0, 2016\n\tglobal_load_lds_dwordx4 %0, off offset:32\n\ts_add_u32 m0, m0, 2016\n\t"
                 "global_load_lds_dwordx4 %0, off offset:64\n\ts_add_u32 m0, m0, 2016\n\tglobal_load_lds_dwordx4 %0, off offset:96\n\ts_add_u32 m0, m0, 2016\n\t"
                 "global_load_lds_dwordx4 %0, off offset:128\n\ts_sub_u32 m0, m0, 7040\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_add_u32 m0, m0, 2016\n\tglobal_load_lds_dwordx4 %1, off offset:32\n\ts_add_u32 m0, m0, 2016\n\t"
                 "global_load_lds_dwordx4 %1, off offset:64\n\ts_add_u32 m0, m0, 2016\n\tglobal_load_lds_dwordx4 %1, off offset:96\n\ts_add_u32 m0, m0, 2016\n\t"
                 "global_load_lds_dwordx4 %1, off offset:128"
                 : : "v"(ge), "v"(go), "s"(lds) : "memory", "m0");
}
// 16 failing-window bits of a piece for windows of 4: bit b <=> bytes b .. b+3 of (q, nx) sum to < thr (nthr = 65536 - thr in every half)
__device__ __forceinline__ uint32_t f6_fail16_w4(const uint4 &q, uint32_t nx, uint64_t nthr) {
    const uint64_t s0 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.x | ((uint64_t)q.y << 32), 0u, nthr);
    const uint64_t s1 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.y | ((uint64_t)q.z << 32), 0u, nthr);
    const uint64_t s2 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.z | ((uint64_t)q.w << 32), 0u, nthr);
    const uint64_t s3 = __builtin_amdgcn_qsad_pk_u16_u8((uint64_t)q.w | ((uint64_t)nx << 32), 0u, nthr);
    // the high bytes of the four sums of an instruction (bit 7 = the sign) side by side, then weighted: flag 0x80 x 2^b = bit 7 + b
    const uint32_t g0 = __builtin_amdgcn_perm((uint32_t)(s0 >> 32), (uint32_t)s0, 0x07050301u) & 0x80808080u;
    const uint32_t g1 = __builtin_amdgcn_perm((uint32_t)(s1 >> 32), (uint32_t)s1, 0x07050301u) & 0x80808080u;
    const uint32_t g2 = __builtin_amdgcn_perm((uint32_t)(s2 >> 32), (uint32_t)s2, 0x07050301u) & 0x80808080u;
    const uint32_t g3 = __builtin_amdgcn_perm((uint32_t)(s3 >> 32), (uint32_t)s3, 0x07050301u) & 0x80808080u;
    uint32_t lo = __builtin_amdgcn_udot4(g0, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(g1, 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(g2, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(g3, 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);                                  // bits 7..14 of lo -> 0..7, of hi -> 8..15
}
template <int W>
__device__ __forceinline__ uint32_t f6_fail16(const uint4 &q, const uint4 &nq, uint32_t thr, uint64_t nthr) {
    if (W == 4) return f6_fail16_w4(q, nq.x, nthr);
    return window_fail_bits16<W>(make_uint2(q.x, q.y), make_uint2(q.z, q.w), thr) |
           (window_fail_bits16<W>(make_uint2(q.z, q.w), make_uint2(nq.x, nq.y), thr) << 8);
}

// byte flags 0x80 "quality >= mq" of four qualities (mq <= 128, mqb = mq in every byte): or, sub, one boolean op
__device__ __forceinline__ uint32_t f6_ok80(uint32_t q, uint32_t mqb) {
    const uint32_t t = (q | 0x80808080u) - mqb;
    return (t | q) & 0x80808080u;
}
// nibble mask of eight bases (the layout of a dword of packed bases: byte j = base 2j << 4 | base 2j + 1) from the byte
// flags o0 (bases 0..3) and o1 (bases 4..7): 0xF where the flag is set
__device__ __forceinline__ uint32_t f6_keep8(uint32_t o0, uint32_t o1) {
    const uint32_t e = __builtin_amdgcn_perm(o1, o0, 0x06040200u);      // flags of the even bases, one per byte
    const uint32_t o = __builtin_amdgcn_perm(o1, o0, 0x07050301u);      // ... of the odd bases
    // 0x80 -> 0xF0: 2^(8j+8) - 2^(8j+4) (the top byte's 2^32 wraps away); 0x80 -> 0x0F: 2^(8j+4) - 2^(8j)
    return ((e << 1) - (e >> 3)) | ((o >> 3) - (o >> 7));
}
// nibble mask of the bases [klo, khi) of a 16-base piece (two dwords of packed bases), klo / khi in 0..16
__device__ __forceinline__ uint2 f6_range_nibbles(int32_t klo, int32_t khi) {
    const bool any = klo < khi;
    const uint64_t ones = ~0ull;
    // linear order (base i at bits 4i .. 4i+3), then the two nibbles of every byte swapped (the first base is the high nibble)
    const uint64_t lin = any ? ((ones << (4 * (klo & 15))) & (ones >> ((64 - 4 * khi) & 63))) : 0ull;
    const uint32_t a = (uint32_t)lin, b = (uint32_t)(lin >> 32);
    return make_uint2(((a & 0x0F0F0F0Fu) << 4) | ((a >> 4) & 0x0F0F0F0Fu), ((b & 0x0F0F0F0Fu) << 4) | ((b >> 4) & 0x0F0F0F0Fu));
}
// the 16 bases of a piece (masked codes m) into the counter words from wb on.  The adds of a piece are ONE asm statement: shift
// counts se / so hold the even / odd bases of a half (byte j = base 2j resp. 2j + 1), `one' << count goes to word wb + 4 * base
#define F6_SH(d, sh, j) "v_lshlrev_b32_sdwa " d ", " sh ", %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #j " src1_sel:DWORD\n\t"
template <int B0>
__device__ __forceinline__ void f6_add8(uint32_t wb, uint32_t se, uint32_t so, uint32_t one) {
    uint32_t t0, t1;
    asm volatile(F6_SH("%0", "%4", 0) F6_SH("%1", "%5", 0)
                 "ds_add_u32 %2, %0 offset:%6\n\t" F6_SH("%0", "%4", 1) "ds_add_u32 %2, %1 offset:%7\n\t" F6_SH("%1", "%5", 1)
                 "ds_add_u32 %2, %0 offset:%8\n\t" F6_SH("%0", "%4", 2) "ds_add_u32 %2, %1 offset:%9\n\t" F6_SH("%1", "%5", 2)
                 "ds_add_u32 %2, %0 offset:%10\n\t" F6_SH("%0", "%4", 3) "ds_add_u32 %2, %1 offset:%11\n\t" F6_SH("%1", "%5", 3)
                 "ds_add_u32 %2, %0 offset:%12\n\tds_add_u32 %2, %1 offset:%13"
                 : "=&v"(t0), "=&v"(t1) : "v"(wb), "v"(one), "v"(se), "v"(so),
                   "n"(4 * B0), "n"(4 * B0 + 4), "n"(4 * B0 + 8), "n"(4 * B0 + 12), "n"(4 * B0 + 16), "n"(4 * B0 + 20), "n"(4 * B0 + 24), "n"(4 * B0 + 28) : "memory");
}
#undef F6_SH
__device__ __forceinline__ void f6_count16(const uint2 &m, uint32_t wb, uint32_t one, uint32_t &badacc) {
    const uint32_t se0 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, (m.x >> 4) & 0x0F0F0F0Fu);      // even bases of the first half
    const uint32_t so0 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, m.x & 0x0F0F0F0Fu);
    const uint32_t se1 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, (m.y >> 4) & 0x0F0F0F0Fu);
    const uint32_t so1 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, m.y & 0x0F0F0F0Fu);
    // codes that are not one of A C G T: bit 7 of the shift count, or code 12 (which the permute turns into a zero like T's)
    badacc |= ((se0 | so0 | se1 | so1) & 0x80808080u) | (((m.x & (m.x >> 1)) | (m.y & (m.y >> 1))) & 0x44444444u);
    if (AMP_F6_ABL & 1) { asm volatile("" : : "v"(wb), "v"(se0 + so0 + se1 + so1)); return; }
    f6_add8<0>(wb, se0, so0, one);
    f6_add8<8>(wb, se1, so1, one);
}

// minimum over the lanes of the wave (DPP row shifts and row broadcasts, no LDS traffic); the result is uniform
__device__ __forceinline__ int32_t wave_min_i32(int32_t x) {
    int32_t t = x, u;
    const int32_t big = 0x7FFFFFFF;
    u = __builtin_amdgcn_update_dpp(big, t, 0x111, 0xF, 0xF, false); t = u < t ? u : t;      // row_shr:1
    u = __builtin_amdgcn_update_dpp(big, t, 0x112, 0xF, 0xF, false); t = u < t ? u : t;      // row_shr:2
    u = __builtin_amdgcn_update_dpp(big, t, 0x114, 0xF, 0xF, false); t = u < t ? u : t;      // row_shr:4
    u = __builtin_amdgcn_update_dpp(big, t, 0x118, 0xF, 0xF, false); t = u < t ? u : t;      // row_shr:8
    u = __builtin_amdgcn_update_dpp(big, t, 0x142, 0xA, 0xF, false); t = u < t ? u : t;      // row_bcast:15 into rows 1 and 3
    u = __builtin_amdgcn_update_dpp(big, t, 0x143, 0xC, 0xF, false); t = u < t ? u : t;      // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(t, 63);
}

// phase stamps of development builds (-DAMP_F6_STAMPS: cycles per phase summed over the waves into ctr[8 ..], turns into ctr[6]); the
// shipped library has none
#ifdef AMP_F6_WAITSTAMPS
#define F6_WS_DECL unsigned long long f6_w[4] = {0, 0, 0, 0}, f6_wt = 0; unsigned long long f6_wn = 0
#define F6_WS_BEGIN do { __builtin_amdgcn_s_waitcnt(0xC07F); f6_wt = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
#define F6_WS_END(k) do { const unsigned long long f6_n = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); f6_w[k] += f6_n - f6_wt; f6_wt = f6_n; } while (0)
#define F6_WS_OUT do { if (lane == 0) { for (int k = 0; k < 4; ++k) atomicAdd(&ctr[8 + k], f6_w[k]); atomicAdd(&ctr[6], f6_wn); } } while (0)
#else
#define F6_WS_DECL
#define F6_WS_BEGIN
#define F6_WS_END(k)
#define F6_WS_OUT
#endif
#ifdef AMP_F6_STAMPS
#define F6_STAMP_DECL unsigned long long f6_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f6_prev = __builtin_amdgcn_s_memtime(); unsigned long long f6_turns = 0
#define F6_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long f6_n = __builtin_amdgcn_s_memtime(); f6_t[k] += f6_n - f6_prev; f6_prev = f6_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define F6_STAMP_OUT do { if (lane == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&ctr[8 + k], f6_t[k]); atomicAdd(&ctr[6], f6_turns); } } while (0)
#define F6_TURN ++f6_turns
#else
#define F6_STAMP_DECL
#define F6_STAMP(k)
#define F6_STAMP_OUT
#define F6_TURN
#endif

struct F6Hdr {                 // a read's header as kept between tiles
    int32_t pos;
    uint32_t lf;               // l_seq (saturated at 0xFFFF) | paired << 16 | reverse << 17 | template-length test of A:452 << 18 | CIGAR ops (saturated at 7) << 19 | valid << 22
    uint32_t c0, o8, idx;
    __device__ uint32_t lseq() const { return lf & 0xFFFFu; }
    __device__ uint32_t nops() const { return (lf >> 19) & 7u; }
    __device__ uint32_t flag() const { return ((lf >> 16) & 1u) | (((lf >> 17) & 1u) << 4); }
    __device__ bool isize_flag() const { return (lf >> 18) & 1u; }
    __device__ bool valid() const { return (lf >> 22) & 1u; }
};

// The arguments are passed one by one, every pointer followed by a 32-bit value: a struct argument (and a run of adjacent
// pointers) is loaded as ONE wide register tuple, and a kernel as short of scalar registers as this one then spills and
// reloads the whole tuple around every use of one field (16 v_readlane for one pointer, several times per tile).
#define F6_PARAMS \
    const int32_t *a_pos, int32_t a_min_quality, const uint16_t *a_flag, int32_t a_window, const int32_t *a_tlen, int32_t a_do_trim, \
    const uint32_t *a_lseq, int32_t a_do_count, const uint32_t *a_cig_off32, int32_t a_ref_len, const uint32_t *a_cig, int32_t a_max_primer_len, \
    const uint32_t *a_seq_off8, int32_t reads_per_block, const uint8_t *a_seq, int32_t a_epoch, const uint8_t *a_qual, int32_t pad1, \
    const int32_t *a_min_start, int32_t pad2, const int32_t *a_max_end, int32_t pad3, \
    int32_t *a_new_pos, int32_t pad4, uint32_t *a_new_ncig, int32_t pad5, uint32_t *a_new_cig, int32_t pad6, int32_t *a_o_ref_len, int32_t pad7, \
    uint8_t *a_trim_flags, int32_t pad8, uint8_t *a_status, int32_t pad9, \
    uint32_t *counts, int32_t pad10, amp_ins_event *a_ev, int32_t pad11, unsigned long long *a_ctr, int32_t pad12, uint32_t *a_ins_at, int32_t pad13, \
    uint32_t *glist, int32_t pad14, uint32_t *gcnt, int32_t pad15, uint32_t *clist, int32_t pad16, \
    int64_t a_n_reads, int32_t pad17, uint64_t read_base, int32_t pad18, long long a_ev_cap

template <int W>
__global__ void __launch_bounds__(F6_WAVES * 64, 2)
k_fast6(F6_PARAMS) {
    const KParams P{a_min_quality, a_window, a_do_trim, a_do_count, a_ref_len, a_max_primer_len, a_min_start, a_max_end, (uint32_t)a_epoch};
    const amp_dev_reads rd{a_n_reads, a_pos, a_flag, a_tlen, a_lseq, a_cig_off32, a_cig, a_seq_off8, a_seq, a_qual, 0, 0};
    const DevOut out{a_new_pos, a_new_ncig, a_new_cig, a_o_ref_len, a_trim_flags, a_status};
    const EventBuf eb{a_ev, a_ctr, a_ins_at, a_ev_cap};
    // separate LDS objects: the compiler only builds alias scopes per LDS variable
    __shared__ uint4 s_q[F6_WAVES][F6_QB / 16];                       // per wave: the tile's qualities, chunk-major
    __shared__ uint4 s_s[F6_WAVES][F6_SB / 16];                       // per wave: the tile's MASKED packed bases, piece-major
    __shared__ uint32_t s_pwin[F6_WAVES][F6_REP * F6_REPW];           // per wave: packed counters
    __shared__ uint32_t s_bwin[F_BPL * F6_BW];                        // the block's window, 32-bit counters: A C G T '-' insertion tally
    __shared__ uint32_t s_ticket, s_gcur, s_n[2];
    unsigned long long *const ctr = eb.ctr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t n = rd.n_reads;
    const int64_t rb = (int64_t)blockIdx.x * reads_per_block;
    const int64_t re = rb + reads_per_block < n ? rb + reads_per_block : n;
    lds_u32 *const bwin = (lds_u32 *)s_bwin;
    lds_u32 *const pwin = (lds_u32 *)s_pwin[wave];
    for (int i = tid; i < F_BPL * F6_BW; i += F6_WAVES * 64) bwin[i] = 0;
    for (int i = lane; i < F6_REP * F6_REPW; i += 64) pwin[i] = 0;
    if (tid == 0) { s_ticket = 0; s_gcur = 0; s_n[0] = 0; s_n[1] = 0; }
    if (tid == 0 && blockIdx.x == 0) { eb.ctr[26] = 0ull; eb.ctr[27] = 0ull; eb.ctr[28] = 0ull; }      // k_gcompact / k_long's counters (amp_wave.hpp)
    int32_t bw_base = rb < n ? rd.pos[rb] : 0;
    bw_base = (bw_base < 16 ? 0 : bw_base - 16) & ~15;
    __syncthreads();

    // ---- classes: the block's reads by the number of their CIGAR ops.  Simple reads fill the block's segment of clist from
    // the front, indel reads from the back; the others go on the general list at once (the order inside a list is the order
    // in which the waves' groups of 64 reads arrive: a tile does not rely on it) ------------------------------------------------
    {
        uint32_t *const seg = clist + rb;
        constexpr int R = 4;                                       // groups of 64 reads whose loads are in flight together
        for (int64_t g0 = rb + (int64_t)wave * 64; g0 < re; g0 += (int64_t)R * F6_WAVES * 64) {
            uint32_t c0[R], c1[R], ls[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t i = g0 + (int64_t)r * F6_WAVES * 64 + lane;
                const int64_t ic = i < re ? i : rb;
                c0[r] = rd.cig_off32[ic]; c1[r] = rd.cig_off32[ic + 1]; ls[r] = rd.lseq[ic];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t i = g0 + (int64_t)r * F6_WAVES * 64 + lane;
                const bool valid = i < re;
                const uint32_t nops = c1[r] - c0[r];
                const bool fits = ls[r] >= 1u && ls[r] <= (uint32_t)F6_MAXLEN;
                const uint32_t cls = !valid ? 3u : (fits && nops >= 1u && nops <= 2u) ? 0u : (fits && nops >= 3u && nops <= 5u) ? 1u : 2u;
#pragma unroll
                for (uint32_t c = 0; c < 3u; ++c) {
                    const unsigned long long m = __ballot(cls == c);
                    if (!m) continue;
                    uint32_t base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add(c == 2u ? (lds_u32 *)&s_gcur : (lds_u32 *)&s_n[c], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (cls == c) {
                        if (c == 0u) seg[at] = (uint32_t)i;
                        else if (c == 1u) seg[(uint32_t)reads_per_block - 1u - at] = (uint32_t)i;
                        else glist[(size_t)rb + at] = (uint32_t)i;
                    }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nS = s_n[0], nI = s_n[1];
    const uint32_t nTS = (nS + 63u) >> 6, nTI = (nI + 63u) >> 6, n_tb = (AMP_F6_ABL & 64) ? 0u : nTS + nTI;

    const int32_t mq = P.min_quality;
    const uint32_t thr = (uint32_t)mq * (uint32_t)W;                // mq <= 128 (the host sends other runs to the general kernel)
    const uint64_t nthr = (uint64_t)((0x10000u - thr) & 0xFFFFu) * 0x0001000100010001ull;
    const uint32_t mqb = (uint32_t)mq * 0x01010101u;
    const uint32_t G = (uint32_t)P.ref_len;
    const uint32_t q_tot8 = rd.seq_off8[n];                          // rows end here (units of 8 bases): 16 bytes of slack behind
    auto take_ticket = [&]() {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((lds_u32 *)&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    unsigned long long n_err = 0;
    // lane constants
    const uint32_t rep = ((uint32_t)lane >> 1) & (uint32_t)(F6_REP - 1);
    const uint32_t wrep = (uint32_t)(uintptr_t)((lds_u8 *)pwin + rep * (uint32_t)(F6_REPW * 4));
    const uint32_t qb_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u8 *)s_q[wave]);
    const lds_u8 *const qrow = (const lds_u8 *)s_q[wave] + 1024 * (lane >> 5) + 32 * (lane & 31);      // + 2048 (p >> 1) + 16 (p & 1): piece p of the lane's row
    lds_u8 *const srow8 = (lds_u8 *)s_s[wave] + 8 * lane;            // + 512 p: masked piece p
    const uint32_t one = 1u;
    int32_t pw_base = 0;
    int pw_tiles = F6_FLUSH;
    uint32_t pw_lim = 0;
    const unsigned ev_shard = blockIdx.x & (EV_SHARDS - 1);
    amp_ins_event *const ev_list = eb.ev + (size_t)ev_shard * (size_t)eb.cap;
    unsigned long long ev_base = 0;
    uint32_t ev_left = 0;

    // folds the wave's packed window into the block's 32-bit window (or the global table) and clears it
    auto fold = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the adds of the counting phase are invisible to the compiler's wait counts)
        wave_sync();
#pragma unroll 1
        for (int idx = lane; idx < F6_PW; idx += 64) {
            uint32_t tc = 0, ag = 0;                                 // T | C << 16, A | G << 16
#pragma unroll
            for (int r = 0; r < F6_REP; ++r) {
                const uint32_t w = pwin[r * F6_REPW + idx] & 0x7FFFFFFFu;
                pwin[r * F6_REPW + idx] = 0;
                tc += w & 0x00FF00FFu; ag += (w >> 8) & 0x00FF00FFu;
            }
            if (tc | ag) {
                const int32_t p = pw_base + idx;
                const uint32_t d = (uint32_t)(p - bw_base);
                const uint32_t c4[4] = {ag & 0xFFFFu, tc >> 16, ag >> 16, tc & 0xFFFFu};      // A C G T
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (!c4[c]) continue;
                    if (d < (uint32_t)F6_BW) lds_add(bwin + c * F6_BW + d, c4[c]);
                    else if ((uint32_t)p < G) atomicAdd(&counts[(size_t)p * AMP_NSYM + c], c4[c]);
                }
            }
        }
        wave_sync();
    };
    auto pad_events = [&]() {
        if ((uint32_t)lane < ev_left && (long long)(ev_base + (unsigned)lane) < eb.cap) ev_list[ev_base + (unsigned)lane] = amp_ins_event{-1, 0u, 0, 0};
    };
    auto set_window = [&](int32_t lead_pos) {
        pw_base = (lead_pos < 16 ? 0 : lead_pos - 16) & ~15;
        pw_lim = (int64_t)G - pw_base >= (int64_t)F6_PW ? (uint32_t)F6_PW : (uint32_t)(G > (uint32_t)pw_base ? G - (uint32_t)pw_base : 0u);
    };

    // ---- the pipeline.  Stage E: a tile's list entries; H: the headers of its reads; C: their CIGAR words and quality rows;
    // T: the two primer-table entries and the rows of packed bases.  While tile t is computed, tile t + 1 has done E, H and C,
    // tile t + 2 E and H, tile t + 3 E.  Loads are requested at two places of a turn (behind pass 1b, behind pass 2) and
    // waited for at the other one, so nothing waits for what it has just requested ------------------------------------------
    struct Raw { int32_t pos, tlen; uint32_t lseq, flag, c0, c1, o8; };
    auto entry_of = [&](uint32_t tk) -> uint32_t {                   // list entry of the lane in tile tk: read index, 0xFFFFFFFF = none
        const bool ind = tk >= nTS;
        const uint32_t j = (ind ? tk - nTS : tk) * 64u + (uint32_t)lane;
        const bool valid = tk < n_tb && j < (ind ? nI : nS);
        const uint32_t at = ind ? (uint32_t)reads_per_block - 1u - j : j;
        const uint32_t e = clist[rb + (valid ? at : 0u)];
        return valid ? e : 0xFFFFFFFFu;
    };
    auto load_raw = [&](uint32_t ent) {
        const int64_t i = ent == 0xFFFFFFFFu ? rb : (int64_t)ent;     // (a lane without a read points at the block's first one)
        Raw h;
        h.pos = rd.pos[i]; h.flag = rd.flag[i]; h.tlen = rd.tlen[i]; h.lseq = rd.lseq[i];
        h.c0 = rd.cig_off32[i]; h.c1 = rd.cig_off32[i + 1]; h.o8 = rd.seq_off8[i];
        return h;
    };
    auto pack_hdr = [&](const Raw &h, uint32_t ent) {
        const uint32_t nn = h.c1 - h.c0, at = (uint32_t)(h.tlen < 0 ? -(int64_t)h.tlen : (int64_t)h.tlen);
        const bool isz = ((int64_t)at - P.max_primer_len) > (int64_t)h.lseq;                                  // A:452
        const bool valid = ent != 0xFFFFFFFFu;
        return F6Hdr{h.pos, (h.lseq > 0xFFFFu ? 0xFFFFu : h.lseq) | ((h.flag & 1u) << 16) | (((h.flag >> 4) & 1u) << 17) | ((isz ? 1u : 0u) << 18) |
                                ((nn > 7u ? 7u : nn) << 19) | ((valid ? 1u : 0u) << 22),
                     h.c0, h.o8, valid ? ent : (uint32_t)rb};
    };
    struct Cg { uint32_t w[5]; };
    auto load_cig = [&](const F6Hdr &h, bool indel_tile) {
        Cg c;
        const uint32_t nops = h.nops();
        c.w[0] = rd.cig[h.c0];
        c.w[1] = rd.cig[h.c0 + (1u < nops ? 1u : 0u)];                               // (words past the read's own repeat its first)
        c.w[2] = c.w[3] = c.w[4] = c.w[0];
        if (indel_tile) {                                                              // (uniform: a tile of simple reads has two ops at most)
#pragma unroll
            for (uint32_t k = 2; k < 5u; ++k) c.w[k] = rd.cig[h.c0 + (k < nops ? k : 0u)];
        }
        return c;
    };
    // quality rows of a tile: instruction s moves chunk 2 (s >> 1) + (lane & 1) of row 32 (s & 1) + (lane >> 1) to 1024 s + 16 lane
    auto issue_q = [&](const F6Hdr &h) {
        const uint32_t o8e = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >> 1) * 4, (int)h.o8);
        const uint32_t o8o = (uint32_t)__builtin_amdgcn_ds_bpermute((32 + (lane >> 1)) * 4, (int)h.o8);
        const uint8_t *qe = rd.qual + (int64_t)o8e * 8 + (lane & 1) * 16, *qo = rd.qual + (int64_t)o8o * 8 + (lane & 1) * 16;
        // a row within 160 bytes of the end of the buffer: its chunks beyond the rows are fetched from where the rows end
        const bool near = (uint64_t)o8e + 20u > (uint64_t)q_tot8 + 2u || (uint64_t)o8o + 20u > (uint64_t)q_tot8 + 2u;
        if (__ballot(near)) {
            const uint32_t lime = (q_tot8 - o8e) * 8u, limo = (q_tot8 - o8o) * 8u;                 // offsets up to here stay inside (16 bytes of slack)
#pragma unroll
            for (int s = 0; s < F6_NP; ++s) {
                const uint32_t want = (uint32_t)(32 * (s >> 1) + 16 * (lane & 1)), lim = (s & 1) ? limo : lime;
                const uint8_t *g = rd.qual + (int64_t)((s & 1) ? o8o : o8e) * 8 + (want < lim ? want : lim);
                dma16_off<0>(g, qb_w + 1024u * (uint32_t)s);
            }
            return;
        }
        if (AMP_F6_ABL & 256) {
            const uint8_t *run = rd.qual + (int64_t)__builtin_amdgcn_readfirstlane((int)h.o8) * 8 + lane * 16;
            dma_q_rows(run, run + 1024, qb_w);          // (timing experiment: ten instructions over one contiguous run; the image is wrong)
            return;
        }
        dma_q_rows(qe, qo, qb_w);
    };
    // rows of packed bases: every lane loads the 80 bytes of its own row into registers (five 16-byte loads).  They are asked for
    // together with the quality rows, half a tile before they are used, and need no staging buffer of their own: the LDS only
    // holds the MASKED bases of the tile that is being counted.  (A row within 80 bytes of the end of the buffer: its chunks
    // behind the rows are fetched from where the rows end.)
    struct SRaw { uint4 c[F6_NP / 2]; };
    auto load_s = [&](const F6Hdr &h) {
        SRaw r;
        const uint8_t *sr = rd.seq + (int64_t)h.o8 * 4;
        const uint32_t lim = (q_tot8 - h.o8) * 4u;                      // offsets up to here stay inside (16 bytes of slack)
#pragma unroll
        for (int c = 0; c < F6_NP / 2; ++c) {
            const uint32_t off = (uint32_t)(16 * c) < lim ? (uint32_t)(16 * c) : lim;
            const uint32_t *g = (const uint32_t *)(sr + off);
            r.c[c] = make_uint4(g[0], g[1], g[2], g[3]);
        }
        return r;
    };
    struct Shape { Bf s; bool ok; int32_t refspan; };
    auto shape_of = [&](const F6Hdr &h, const Cg &c, bool indel_tile) {
        Shape r;
        bool ok;
        r.s = bf_from_words5((int)h.nops(), c.w[0], c.w[1], c.w[2], c.w[3], c.w[4], (int32_t)h.lseq(), F_MAXINS, F_MAXDEL, ok);
        r.ok = ok & h.valid() & (indel_tile | (r.s.kind == 0));
        r.refspan = r.ok ? r.s.m1 + r.s.m2 + r.s.kD() : 1;
        return r;
    };
    struct Tabs { int32_t L, R; };
    auto load_tabs = [&](const F6Hdr &h, const Shape &sh) {
        Tabs t{-1, -1};
        const bool in_ref = (uint32_t)h.pos < G && (uint32_t)(h.pos + sh.refspan - 1) < G;
        if (sh.ok && P.do_trim && in_ref) { t.L = P.max_end[h.pos]; t.R = P.min_start[h.pos + sh.refspan - 1]; }
        return t;
    };
    // a read's results (stored behind the trims: pass 1b and pass 2 lie between the stores and the next wait for loads)
    auto store_results = [&](uint32_t i, uint32_t slot_lo, uint32_t ncig, const uint32_t (&cw)[5], int32_t npos, int32_t reflen, uint32_t flags, uint32_t status) {
        uint32_t *home = out.new_cig + ((size_t)slot_lo + 3 * (size_t)i);
        if (ncig > 0u) home[0] = cw[0];
        if (ncig > 1u) home[1] = cw[1];
        if (ncig > 2u) home[2] = cw[2];
        if (ncig > 3u) home[3] = cw[3];
        if (ncig > 4u) home[4] = cw[4];
        if (out.new_pos) out.new_pos[i] = npos;
        if (out.new_ncig) out.new_ncig[i] = ncig;
        if (out.ref_len) out.ref_len[i] = reflen;
        if (out.trim_flags) out.trim_flags[i] = (uint8_t)flags;
        if (out.status) out.status[i] = (uint8_t)status;
    };
    // hand-over to the general pass: the block's segment of the list
    auto push_list = [&](bool has, uint32_t entry) {
        const unsigned long long m = __ballot(has);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add((lds_u32 *)&s_gcur, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (has) glist[(size_t)rb + base + __popcll(m & ((1ull << lane) - 1ull))] = entry;
        }
    };

#ifndef AMP_F6_STAGGER
#define AMP_F6_STAGGER 4000
#endif
    if (AMP_F6_STAGGER > 0) {
        // the waves of a block (and the blocks of a launch) start together and take equally long per tile: left alone they ask
        // for their rows at the same moments and compute at the same moments, so the memory system and the vector pipes take
        // turns instead of working side by side.  Wave w starts w steps late
        const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)AMP_F6_STAGGER * (unsigned)wave;
        while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(8);
    }
    // ---- prologue of the pipeline.  Across the back edge of the loop only values that have ARRIVED are carried (the two kinds of
    // turn keep them in different registers: the compiler copies them there, and a copy of a register with a load in flight is
    // answered with a full wait) ---------------------------------------------------------------------------------------------
    uint32_t tk0 = take_ticket(), tk1 = take_ticket(), tk2 = take_ticket();
    uint32_t e2;
    F6Hdr h0, h1;
    Cg cw0;
    SRaw sr0;
    {
        const uint32_t e0 = entry_of(tk0), e1 = entry_of(tk1);
        e2 = entry_of(tk2);
        h0 = pack_hdr(load_raw(e0), e0);
        h1 = pack_hdr(load_raw(e1), e1);
        cw0 = load_cig(h0, tk0 >= nTS);
        sr0 = load_s(h0);
        if (tk0 < n_tb) issue_q(h0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
    }

    F6_STAMP_DECL;
    F6_WS_DECL;
    auto turn = [&](auto itag) {
        constexpr bool ITILE = decltype(itag)::value;                   // a tile of indel reads
        F6_TURN; F6_STAMP(7);
        uint32_t tk3v = 0;                           // (the ticket of the tile three ahead: asked for now, read behind pass 1b)
        if (lane == 0) tk3v = __hip_atomic_fetch_add((lds_u32 *)&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const F6Hdr h = h0;
        // ---- requests, first half: this tile's primer-table entries, the header of the tile after the next
        Shape shp = shape_of(h, cw0, ITILE);
        if (!ITILE) { shp.s.kind = 0; shp.s.k = 0; shp.s.m2 = 0; }      // (constants for the compiler: the closed forms fold)
        const Tabs tA = load_tabs(h, shp);
        const Raw r2 = load_raw(e2);
        const int64_t i = (int64_t)h.idx;
        const int32_t pos = h.pos;
        const uint32_t lseq = h.lseq(), flag = h.flag();
        const bool rev = (flag & 0x10u) != 0;
        // ---- the wave's packed window: fold and re-anchor when the tile has moved on, or before a byte could overflow.
        // The tile's leftmost read anchors it (the order inside a list is almost, not exactly, the batch's)
        int32_t minpos;
        {
            minpos = wave_min_i32(h.valid() ? pos : 0x7FFFFFFF);
            const int32_t want = (minpos < 16 ? 0 : minpos - 16) & ~15;
            if (pw_tiles >= F6_FLUSH || want < pw_base || want - pw_base >= 32) {
                if (pw_tiles) fold();
                set_window(minpos); pw_tiles = 0;
            }
            ++pw_tiles;
        }
        F6_STAMP(0);          // requests, window anchor
        // ---- pass 1a: failing windows of the whole read, pieces in their natural order (the quality rows were waited for at
        // the end of the previous turn) --------------------------------------------------------------------------------------
        uint32_t F[F6_NP / 2];
        uint32_t fb;
        {
            auto piece = [&](int k) -> uint4 { const amp_u32x4 a = *(const lds_u32x4 *)(qrow + 2048 * (k >> 1) + 16 * (k & 1)); return make_uint4(a.x, a.y, a.z, a.w); };
            uint4 q0 = piece(0), q1 = piece(1), q2 = piece(2);          // (three pieces on their way: a round trip to the LDS takes longer than a piece's arithmetic)
            fb = q0.x & 0xFFu;
#pragma unroll
            for (int k = 0; k < F6_NP; ++k) {
                uint4 q3 = q2;
                if (k + 3 < F6_NP) q3 = piece(k + 3);
                const uint32_t f16 = (P.do_trim && !(AMP_F6_ABL & 4)) ? f6_fail16<W>(q0, q1, thr, nthr) : (q0.x & 1u);
                if (k & 1) F[k >> 1] |= f16 << 16; else F[k >> 1] = f16;
                q0 = q1; q1 = q2; q2 = q3;
                __builtin_amdgcn_sched_barrier(0);          // one piece at a time: interleaving them costs registers
            }
        }
        F6_STAMP(1);          // pass 1a
        // (no wait here: the clips below wait for the two table entries only -- they are the oldest requests of the turn -- and run while
        //  the rows of packed bases are still on their way)
        // ---- primer clips in closed form (A:450-558) ---------------------------------------------------------------------
        Bf s = shp.s;
        const bool shaped = shp.ok;
        const bool in_ref = (uint32_t)pos < G && (uint32_t)(pos + shp.refspan - 1) < G;          // A:450-451
        const int32_t q_ins = s.kind ? s.a + s.m1 : 0;                 // query index of the first inserted base / of the base behind the deletion
        TrimState ts{pos, 1, 0u, 0};
        {
            const bool trim = shaped & (P.do_trim != 0) & !(AMP_F6_ABL & 128), use = trim & in_ref;
            ts.err = (trim & !in_ref) ? AMP_RS_INDEX_REF : 0;
            int32_t p2 = pos; uint32_t f2 = 0u;
            const Bf sp = bf_trim_primers(s, p2, f2, flag, h.isize_flag(), (int32_t)lseq, tA.L, tA.R);
            s = bf_pick(use, sp, s); ts.pos = use ? p2 : pos; ts.flags = use ? f2 : 0u;
        }
        const bool scan = shaped & (P.do_trim != 0) & (ts.err == 0) & !s.punt & !(AMP_F6_ABL & 128);
        int32_t lo, qlen;
        bf_quality_window(s, (int32_t)lseq, lo, qlen);
        lo = scan ? lo : 0; qlen = scan ? qlen : 0;
        const int32_t hi = lo + qlen;
        // ---- first failing window start / last failing window end among the window starts [lo, hi - W] ----------------------
        int32_t ffmin = 0x7FFFFFFF, lemax = -1;
        {
            const int32_t we = hi - W + 1;                              // one past the last window start
#pragma unroll
            for (int w = 0; w < F6_NP / 2; ++w) {
                int32_t a = lo - 32 * w, b = we - 32 * w;
                a = a < 0 ? 0 : a; b = b > 32 ? 32 : b;
                const uint32_t m = (a < b) ? ((0xFFFFFFFFu << a) & (0xFFFFFFFFu >> (32 - b))) : 0u;
                const uint32_t f = F[w] & m;
                const int32_t f1 = 32 * w + (__builtin_ffs((int)f) - 1), e1_ = 32 * w + (31 - __builtin_clz(f)) + W;
                ffmin = (f && f1 < ffmin) ? f1 : ffmin;
                lemax = (f && e1_ > lemax) ? e1_ : lemax;
            }
        }
        // the 3' end's shrinking windows (A:575-576, A:637-638) need the W - 1 qualities at that end; reads with an insertion the
        // qualities of the inserted bases: 16 bytes from an 8-aligned offset of the row each
        auto row16 = [&](int32_t off8) -> uint4 {                      // bytes [off8, off8 + 16) of the lane's quality row, off8 a multiple of 8 (what lies behind the row's 160 bytes: anything)
            const int32_t pc = off8 >> 4, pn = (off8 + 8) >> 4 < F6_NP ? (off8 + 8) >> 4 : F6_NP - 1;
            const amp_u32x2 a = *(const lds_u32x2 *)(qrow + 2048 * (pc >> 1) + 16 * (pc & 1) + (off8 & 8));
            const amp_u32x2 b = *(const lds_u32x2 *)(qrow + 2048 * (pn >> 1) + 16 * (pn & 1) + ((off8 + 8) & 8));
            return make_uint4(a.x, a.y, b.x, b.y);
        };
        const int32_t first = !scan ? 0 : ((rev || qlen < W) ? lo : lo + qlen - W + 1);
        const int32_t tab = first & ~7;
        const uint4 tw = row16(tab);
        // ---- quality clip, results (A:589-686) ------------------------------------------------------------------------------
        bool general = h.valid() && !shaped;
        bool stored = false;
        uint32_t ncig = 0, cw[5] = {0u, 0u, 0u, 0u, 0u};
        int32_t reflen = 0;
        if (shaped) {
            if (fb == 0xFFu || s.punt) {
                general = true;                   // QUAL '*': the generic code reports it (A:561-562, A:718); a shape the closed forms leave
            } else {
                if (scan) {
                    int32_t iq;
                    if (!rev && ffmin != 0x7FFFFFFF) iq = ffmin - lo;
                    else if (rev && lemax >= 0) iq = lemax - lo;
                    else {
                        iq = rev ? 0 : qlen;
                        int32_t acc = 0;
                        const uint64_t t_lo = (uint64_t)tw.x | ((uint64_t)tw.y << 32), t_hi = (uint64_t)tw.z | ((uint64_t)tw.w << 32);
                        const int32_t kmax = qlen < W - 1 ? qlen : W - 1;
                        for (int32_t k = 1; k <= kmax; ++k) {
                            const uint32_t o = (uint32_t)((rev ? lo + k - 1 : hi - k) - tab);      // 0..15
                            acc += (int32_t)(((o < 8u ? t_lo : t_hi) >> ((o & 7u) * 8u)) & 0xFFu);
                            if ((int64_t)acc < (int64_t)mq * k) iq = rev ? k : qlen - k;
                        }
                    }
                    uint32_t f2 = ts.flags;
                    s = bf_trim_quality(s, ts.pos, f2, rev, iq, qlen);
                    ts.flags = f2;
                }
                if (s.punt) {
                    general = true;
                } else {
                    if (!ts.err) {
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
        bool counted = stored && !ts.err && P.do_count;
        // ---- counted query ranges [qa1, qb1) and [qa2, qb2), reference position of their first base -----------------------------
        const bool two = counted && s.kind != 0;
        const int32_t qa1 = counted ? s.a : 0, qb1 = counted ? qa1 + s.m1 : 0;
        const int32_t qa2 = two ? qb1 + s.kI() : qb1, qb2 = two ? qa2 + s.m2 : qa2;
        const int32_t pos2 = ts.pos + s.m1 + s.kD();
        const int32_t end_pos = two ? pos2 + s.m2 : ts.pos + s.m1;       // one past the last counted position
        // without trimming nothing has looked at the reference's end yet (A:753 raises there, but only for a base of good quality):
        // such a read is counted base by base
        bool want_status = false;
        bool slow_all = counted && (uint32_t)end_pos > G;
        F6_STAMP(3);          // clips, results
        F6_WS_BEGIN;
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the header of the tile after the next, requested at the top of the turn
        F6_WS_END(0);
        F6_STAMP(2);          // wait
        // ---- pass 1b: the codes of the low-quality bases become zero; the masked pieces go back piece-major -------------------
        uint4 iq16 = make_uint4(0u, 0u, 0u, 0u);
        const int32_t g_ins = q_ins & ~7;
        if (ITILE) iq16 = row16(g_ins);
        uint32_t zacc = 0;                           // a code 0 ('=') under a good quality: bit 3 of some nibble (pass 2 cannot tell it from a masked base)
        {
            struct Ch { amp_u32x4 qa, qb, sq; };
            auto chunk = [&](int c) { return Ch{*(const lds_u32x4 *)(qrow + 2048 * c), *(const lds_u32x4 *)(qrow + 2048 * c + 16), amp_u32x4{sr0.c[c].x, sr0.c[c].y, sr0.c[c].z, sr0.c[c].w}}; };
            Ch c0 = chunk(0), c1 = chunk(1);                      // (two chunks on their way while one is worked on)
#pragma unroll
            for (int c = 0; c < F6_NP / 2; ++c) {
                Ch c2 = c1;
                if (c + 2 < F6_NP / 2) c2 = chunk(c + 2);
                const amp_u32x4 qa = c0.qa, qb = c0.qb, sq = c0.sq;
                const uint32_t k0 = (AMP_F6_ABL & 8) ? qa.x : f6_keep8(f6_ok80(qa.x, mqb), f6_ok80(qa.y, mqb)), k1 = (AMP_F6_ABL & 8) ? qa.z : f6_keep8(f6_ok80(qa.z, mqb), f6_ok80(qa.w, mqb));
                const uint32_t k2 = (AMP_F6_ABL & 8) ? qb.x : f6_keep8(f6_ok80(qb.x, mqb), f6_ok80(qb.y, mqb)), k3 = (AMP_F6_ABL & 8) ? qb.z : f6_keep8(f6_ok80(qb.z, mqb), f6_ok80(qb.w, mqb));
                { const uint32_t w0 = sq.x | ~k0, w1 = sq.y | ~k1, w2 = sq.z | ~k2, w3 = sq.w | ~k3;      // has-zero-nibble over the kept codes
                  zacc |= ((w0 - 0x11111111u) & ~w0) | ((w1 - 0x11111111u) & ~w1) | ((w2 - 0x11111111u) & ~w2) | ((w3 - 0x11111111u) & ~w3); }
                // (the reads of the chunks ahead are in front of these writes in the queue; a chunk's writes stay inside its own 1 KB)
                *(lds_u32x2 *)(srow8 + 1024 * c) = amp_u32x2{sq.x & k0, sq.y & k1};
                *(lds_u32x2 *)(srow8 + 1024 * c + 512) = amp_u32x2{sq.z & k2, sq.w & k3};
                c0 = c1; c1 = c2;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        zacc &= 0x88888888u;
        wave_sync();                                 // both buffers have been read: the quality image is free
        F6_STAMP(4);          // pass 1b
        // the read's results (pass 2 lies between these stores and the next wait for loads)
        if (stored && !(AMP_F6_ABL & 32)) store_results((uint32_t)i, h.c0, ncig, cw, ts.pos, reflen, ts.err ? 0u : ts.flags, (uint32_t)ts.err);
        const F6Hdr h2 = pack_hdr(r2, e2);          // (the header of the tile after the next has arrived: packed, its seven registers are free)
        // ---- the ends of the counted ranges: the pieces that hold them are masked to the range ----------------------------------
        int32_t jb = 0; uint2 bsq = make_uint2(0u, 0u);                 // (indel tiles) the part of the second range that shares a piece with the first
        bool has_b = false;
        {
            const bool any1 = qb1 > qa1, any2 = qb2 > qa2;
            auto mask_piece = [&](int32_t pc, bool on) {
                lds_u32x2 *w = (lds_u32x2 *)(srow8 + 512 * (on ? pc : 0));
                const amp_u32x2 x = *w;
                const int32_t j0 = 16 * pc;
                uint2 k = f6_range_nibbles(qa1 - j0 < 0 ? 0 : (qa1 - j0 > 16 ? 16 : qa1 - j0), qb1 - j0 > 16 ? 16 : (qb1 - j0 < 0 ? 0 : qb1 - j0));
                if (ITILE) {
                    const uint2 k2 = f6_range_nibbles(qa2 - j0 < 0 ? 0 : (qa2 - j0 > 16 ? 16 : qa2 - j0), qb2 - j0 > 16 ? 16 : (qb2 - j0 < 0 ? 0 : qb2 - j0));
                    k.x |= k2.x; k.y |= k2.y;
                }
                if (on) *w = amp_u32x2{x.x & k.x, x.y & k.y};
            };
            if (ITILE) { mask_piece(qa1 >> 4, any1); mask_piece((qb1 - 1) >> 4, any1); }
            else {
                // (both pieces read before either is written: two round trips become one; the same piece twice gets the same mask twice)
                const int32_t pa = any1 ? qa1 >> 4 : 0, pb = any1 ? (qb1 - 1) >> 4 : 0;
                lds_u32x2 *wa = (lds_u32x2 *)(srow8 + 512 * pa), *wb2 = (lds_u32x2 *)(srow8 + 512 * pb);
                const amp_u32x2 xa = *wa, xb = *wb2;
                const int32_t ja = 16 * pa, jb2 = 16 * pb;
                const uint2 ka = f6_range_nibbles(qa1 - ja < 0 ? 0 : qa1 - ja, qb1 - ja > 16 ? 16 : qb1 - ja);
                const uint2 kb2 = f6_range_nibbles(qa1 - jb2 < 0 ? 0 : qa1 - jb2, qb1 - jb2 > 16 ? 16 : qb1 - jb2);
                if (any1) { *wa = amp_u32x2{xa.x & ka.x, xa.y & ka.y}; *wb2 = amp_u32x2{xb.x & kb2.x, xb.y & kb2.y}; }
            }
            if (ITILE) {
                mask_piece(qa2 >> 4, any2);
                mask_piece((qb2 - 1) >> 4, any2);
                // the piece that holds the end of the first range and the start of the second: its second part is counted from a copy
                const int32_t pj = (qb1 - 1) >> 4;
                has_b = two && any1 && any2 && qa2 < 16 * pj + 16;
                jb = 16 * pj;
                lds_u32x2 *w = (lds_u32x2 *)(srow8 + 512 * (has_b ? pj : 0));
                const amp_u32x2 x = *w;
                const uint2 kb = f6_range_nibbles(qa2 - jb < 0 ? 0 : (qa2 - jb > 16 ? 16 : qa2 - jb), qb2 - jb > 16 ? 16 : (qb2 - jb < 0 ? 0 : qb2 - jb));
                bsq = make_uint2(x.x & kb.x, x.y & kb.y);
                if (has_b) *w = amp_u32x2{x.x & ~kb.x, x.y & ~kb.y};
            }
        }
        bool bad_extra = false;
        if (ITILE) {
            // deletion: '-' at each of its positions (A:714-715), through the block's window
            if (two && s.kind == 2) {
                for (int32_t j = 0; j < s.k; ++j) {
                    const int32_t r = ts.pos + s.m1 + j;
                    const uint32_t d = (uint32_t)(r - bw_base);
                    if ((uint32_t)r >= G) bad_extra = true;
                    else if (d < (uint32_t)F6_BW) lds_add_nt(bwin + 4 * F6_BW + d, 1u);
                    else atomicAdd(&counts[(size_t)r * AMP_NSYM + 5], 1u);
                }
            }
            // insertion (A:730-748): one event per maximal run of good-quality inserted bases (see amp_fast.hpp)
            uint32_t good = 0;
            if (two && s.kind == 1) {
                const uint32_t m16 = ok_bits16(iq16, mqb);
                good = (m16 >> (uint32_t)(s.a + s.m1 - g_ins)) & ((1u << s.k) - 1u);
            }
            uint32_t runs = good & ~(good << 1);
            const unsigned long long em = __ballot(runs != 0u);
            if (em) {
                const uint32_t total = (uint32_t)__popcll(em);
                if (total > ev_left) {
                    pad_events();
                    unsigned long long nb = 0;
                    if (lane == 0) nb = atomicAdd(&ctr[16 + ev_shard], (unsigned long long)F_EVGRAN);
                    ev_base = __shfl(nb, 0); ev_left = F_EVGRAN;
                }
                if (runs) {
                    const int32_t q0 = s.a + s.m1, r2_ = ts.pos + s.m1, ref_end = ts.pos + s.m1 + s.m2;
                    const unsigned long long slot = ev_base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
                    const uint32_t rid = (uint32_t)(read_base + (uint64_t)i);
                    bool firstrun = true;
                    while (runs) {
                        const int32_t js = __builtin_ctz(runs);
                        runs &= runs - 1u;
                        const int32_t je = js + __builtin_ctz(~(good >> js));
                        int32_t elo, ehi;
                        if (je == s.k && s.m2 > 0 && r2_ == 0) py_slice(q0 + js, q0 + je + 1, (int32_t)lseq, elo, ehi);   // A:735-736
                        else py_slice(q0 + js - 1, q0 + je, (int32_t)lseq, elo, ehi);              // A:738
                        int32_t ins_pos = je == s.k ? r2_ : ref_end;                               // A:742 / A:739-740
                        ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;                               // A:744
                        const bool inside = (uint32_t)ins_pos < G;
                        if (!inside) bad_extra = true;
                        if (firstrun) {
                            if ((long long)slot < eb.cap) ev_list[slot] = inside ? amp_ins_event{ins_pos, rid, elo, ehi} : amp_ins_event{-1, 0u, 0, 0};
                            if (inside) {
                                const uint32_t d = (uint32_t)(ins_pos - bw_base);
                                if (d < (uint32_t)F6_BW) lds_add_nt(bwin + 5 * F6_BW + d, 1u);
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
        wave_sync();
        // ---- requests, second half: the next tile's CIGAR words, quality rows (the buffer is free since pass 1b) and packed bases,
        // the list entries of the tile three ahead -- asked for here, behind the register-hungry part of the turn; pass 2 covers the wait
        const uint32_t tk3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk3v);
        const Cg c1w = load_cig(h1, tk1 >= nTS);
        if (tk1 < n_tb && !(AMP_F6_ABL & 16)) issue_q(h1);
        const uint32_t e3 = entry_of(tk3);
        SRaw sr1 = sr0;
        if (!(AMP_F6_ABL & 16)) sr1 = load_s(h1);
        F6_STAMP(5);          // requests, range ends, (indel tiles: deletions, events)
        // ---- pass 2: the masked codes into the wave's packed window.  A tile is counted in passes: every pass takes the lanes
        // whose counted positions lie inside the window, then the window is folded and anchored at the leftmost lane left.
        // Lane l works on piece (k + l) mod np in step k: the lanes of a pile of reads do not add to one address at a time -------
        const uint32_t np = (lseq + 15u) >> 4;
        const uint32_t npc = np < 1u ? 1u : (np > (uint32_t)F6_NP ? (uint32_t)F6_NP : np);
        const uint32_t rot = (uint32_t)lane - npc * (((uint32_t)lane * ((1023u + npc) / npc)) >> 10);      // lane % npc (exact for lane < 64, npc <= 10)
        uint32_t badacc = 0;
        // (slow_all: also a read no window position takes -- the first / last 16 positions of the reference)
        bool todo = counted && !slow_all && !(AMP_F6_ABL & 2);
        for (bool first_pass = true;; first_pass = false) {
            const unsigned long long tm = __ballot(todo);
            if (!tm) break;
            // (the leftmost of the lanes that are left)
            const int32_t lead_pos = wave_min_i32(todo ? ts.pos : 0x7FFFFFFF);
            if (!first_pass || lead_pos - pw_base < 16) {
                if (!first_pass || pw_tiles > 1) fold();
                set_window(lead_pos); pw_tiles = 1;
            }
            const bool fits = ts.pos - pw_base >= 16 && end_pos - pw_base + 16 <= (int32_t)pw_lim;
            const bool lead = todo && ts.pos == lead_pos;
            if (lead && !fits) slow_all = true;                      // (anchored at its own position and still outside)
            const bool now = todo && fits;
            const int32_t dbase1 = ts.pos - pw_base - qa1, dbase2 = pos2 - pw_base - qa2;
            const int32_t a1 = now ? qa1 : 0, b1 = now ? qb1 : 0, a2 = now ? qa2 : 0, b2 = now ? qb2 : 0;
            if (ITILE && __ballot(has_b && now)) {
                int32_t d0 = dbase2 + jb;
                d0 = d0 < 0 ? 0 : (d0 > F6_PW - 16 ? F6_PW - 16 : d0);
                const uint2 mb = (has_b && now) ? bsq : make_uint2(0u, 0u);
                f6_count16(mb, wrep + (uint32_t)d0 * 4u, one, badacc);
            }
            // All ten pieces are read BEFORE the first add: the adds are issued by inline assembly, the compiler does not count them,
            // and its wait for a read that was issued behind a piece's adds would drain those adds too (the LDS returns in order) --
            // the wave would stand still until its own 16 atomics have executed, ten times per tile
            auto piece_of = [&](int k) -> uint32_t { uint32_t p = (uint32_t)k + rot; p = p >= npc ? p - npc : p; return (uint32_t)k < npc ? p : 0u; };
            amp_u32x2 xs[F6_NP];
#pragma unroll
            for (int k = 0; k < F6_NP; ++k) xs[k] = *(const lds_u32x2 *)(srow8 + 512 * piece_of(k));
#pragma unroll
            for (int k = 0; k < F6_NP; ++k) {
                const uint32_t p = piece_of(k);
                const bool slot = (uint32_t)k < npc;
                const int32_t j0 = (int32_t)(p * 16u);
                const amp_u32x2 x = xs[k];
                const bool second = ITILE && (j0 >= b1 || b1 <= a1);      // a piece behind the first range belongs to the second
                const bool in = slot && (second ? (j0 < b2 && j0 + 16 > a2 && a2 < b2) : (j0 < b1 && j0 + 16 > a1 && a1 < b1));
                int32_t d0 = (second ? dbase2 : dbase1) + j0;
                d0 = d0 < 0 ? 0 : (d0 > F6_PW - 16 ? F6_PW - 16 : d0);
                const uint2 m = in ? make_uint2(x.x, x.y) : make_uint2(0u, 0u);
                f6_count16(m, wrep + (uint32_t)d0 * 4u, one, badacc);
                __builtin_amdgcn_sched_barrier(0);
            }
            todo = todo && !now && !slow_all;
        }
        // (the base image was read before the first add: free.  The adds drain while the next turn starts)
        F6_STAMP(6);          // pass 2
        // ---- careful loop (rare): the reads with a counted code outside A C G T (N calls are counted here, anything else wants
        // its exact status), and the reads no window took --------------------------------------------------------------------------
        want_status = want_status || bad_extra;
        if (__ballot((badacc != 0u || zacc != 0u || slow_all) && counted)) {
            if ((badacc != 0u || zacc != 0u || slow_all) && counted) {
                const uint8_t *qrow_g = rd.qual + (int64_t)h.o8 * 8, *srow_g = rd.seq + (int64_t)h.o8 * 4;
                auto careful = [&](int32_t x0, int32_t x1, int32_t rp0) {
                    for (int32_t q = x0; q < x1; ++q) {
                        if ((int32_t)qrow_g[q] < mq) continue;
                        const uint32_t sb = srow_g[q >> 1];
                        const uint32_t col = col_of_code((q & 1) ? (sb & 15u) : (sb >> 4));
                        const int32_t rp = rp0 + (q - x0);
                        if (col < (uint32_t)F_NPL && !slow_all) continue;                     // (counted by pass 2)
                        const uint32_t d = (uint32_t)(rp - bw_base);
                        if (col > 4u || (uint32_t)rp >= G) want_status = true;
                        else if (d < (uint32_t)F6_BW && col < (uint32_t)F_NPL) lds_add_nt(bwin + col * F6_BW + d, 1u);
                        else atomicAdd(&counts[(size_t)rp * AMP_NSYM + col], 1u);
                    }
                };
                careful(qa1, qb1, ts.pos);
                if (two) careful(qa2, qb2, pos2);
            }
        }
        // ---- hand-over to the general pass -------------------------------------------------------------------------------------
        push_list(general || (stored && !ts.err && P.do_count && want_status),
                  (uint32_t)i | (general ? 0u : GL_STATUS_ONLY));          // (status only: a base could not be counted, exact status wanted)
        // ---- everything asked for behind pass 1b has arrived ----------------------------------------------------------------------
#ifdef AMP_F6_WAITSTAMPS
        { const unsigned long long f6_a = __builtin_amdgcn_s_memtime(); F6_WS_BEGIN; f6_w[2] += f6_wt - f6_a; ++f6_wn; }      // (the drain of the adds)
#endif
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        F6_WS_END(1);
        tk0 = tk1; tk1 = tk2; tk2 = tk3;
        h0 = h1; h1 = h2; cw0 = c1w; e2 = e3; sr0 = sr1;
    };
    while (tk0 < n_tb) {
        if (tk0 >= nTS) turn(std::true_type{}); else turn(std::false_type{});
    }
    pad_events();
    if (n_tb) fold();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (asm adds into the block's window: deletions, insertion tally, careful loop)
    __syncthreads();
    for (int i = tid; i < F_BPL * F6_BW; i += F6_WAVES * 64) {
        const uint32_t v = bwin[i];
        if (v) {
            const int pl = i / F6_BW, d = i - pl * F6_BW;
            const uint32_t p = (uint32_t)(bw_base + d);
            if (p < G) {
                if (pl < F_NPL) atomicAdd(&counts[(size_t)p * AMP_NSYM + pl], v);
                else if (pl == 4) atomicAdd(&counts[(size_t)p * AMP_NSYM + 5], v);      // '-'
                else atomicAdd(&eb.ins_at[p], v);
            }
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
    F6_STAMP_OUT;
    F6_WS_OUT;
    if (tid == 0) { gcnt[blockIdx.x] = s_gcur; if (s_gcur) eb.ctr[29] = (unsigned long long)P.epoch; }      // (every block writes the same value)
}

static inline FastGrid fast6_grid(int64_t n_reads, int n_cu) {
    int64_t rpb = (n_reads + (int64_t)n_cu - 1) / (int64_t)n_cu;
    rpb = ((rpb + 63) / 64) * 64;
    if (rpb < 2 * F6_WAVES * 64) rpb = 2 * F6_WAVES * 64;
    return FastGrid{(n_reads + rpb - 1) / rpb, rpb};
}

static inline int fast6_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out, uint32_t *counts,
                               const EventBuf &eb, uint32_t *glist, uint32_t *gcnt, uint32_t *clist, const FastGrid &fg, hipStream_t stream) {
    const unsigned g = (unsigned)fg.grid, t = F6_WAVES * 64;
    const int rpb = (int)fg.rpb;
#define F6_GO(w) k_fast6<w><<<g, t, 0, stream>>>(rd.pos, P.min_quality, rd.flag, P.window, rd.tlen, P.do_trim, rd.lseq, P.do_count, rd.cig_off32, P.ref_len, rd.cig, \
        P.max_primer_len, rd.seq_off8, rpb, rd.seq, (int32_t)P.epoch, rd.qual, 0, P.min_start, 0, P.max_end, 0, out.new_pos, 0, out.new_ncig, 0, out.new_cig, 0, out.ref_len, 0, \
        out.trim_flags, 0, out.status, 0, counts, 0, eb.ev, 0, eb.ctr, 0, eb.ins_at, 0, glist, 0, gcnt, 0, clist, 0, rd.n_reads, 0, read_base, 0, eb.cap)
    switch (P.window) {
        case 1: F6_GO(1); break;
        case 2: F6_GO(2); break;
        case 3: F6_GO(3); break;
        case 4: F6_GO(4); break;
        case 5: F6_GO(5); break;
        case 6: F6_GO(6); break;
        case 7: F6_GO(7); break;
        default: F6_GO(8); break;
    }
#undef F6_GO
    return (int)hipGetLastError();
}

}  // namespace amp
