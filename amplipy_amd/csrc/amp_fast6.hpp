// amp_fast6.hpp -- the fast kernel, third generation (variant 6): trim + pileup of the reads whose CIGAR has the shape
// [S a][M m1]([I|D k][M m2])[S c] and at most 160 bases, one lane per read, every byte of the batch loaded once.  CDNA4 / gfx950.
//
// What the counters of the first two generations showed (profiles/r03_*, tools/micro/valu_rate3.hip): the kernel is bound by
// the vector pipe of a SIMD, and not every instruction costs the same there -- add / sub / shift / and / or / compare /
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
//     The quality buffer is free after 1b and the base buffer after pass 2; the next tile's rows are requested then.
//   * The wave's packed window (one 32-bit word per reference position: T A C G counters of 8 8 8 7 bits + the bit the
//     masked bases hit, 4 replicas) is folded into the block's 32-bit window every 7 tiles at the latest.
// Results (new CIGAR, position, flags, counts, insertion events) are bit-identical to the other variants: same closed
// forms (amp_bf.hpp, fuzzed against the generic code on the CPU), same hand-over list for the general pass.
#pragma once

#include <type_traits>

#include "amp_fast5.hpp"

namespace amp {

constexpr int F6_WAVES = 8;
constexpr int F6_NP = 10;                  // 16-base pieces of a row
constexpr int F6_MAXLEN = 16 * F6_NP;      // longest read taken
constexpr int F6_QB = 1024 * F6_NP;        // bytes of a wave's quality image (10 instructions x 64 lanes x 16 bytes)
constexpr int F6_SB = 512 * F6_NP;         // ... of its base image (5 instructions)
constexpr int F6_PW = 224;                 // reference positions covered by a wave's packed window
constexpr int F6_REP = 4;                  // replicas of it (replica r is skewed by r banks)
constexpr int F6_REPW = F6_PW + 1;
constexpr int F6_BW = 480;                 // positions of the block's 32-bit window
constexpr int F6_FLUSH = 7;                // tiles between two folds: a counter gets at most 64 / F6_REP = 16 increments per tile, G has 7 bits
constexpr uint32_t F6_LUT_LO = 0xFF10081Fu, F6_LUT_HI = 0xFFFFFF18u;      // shift count by code: 0 -> 31, A -> 8, C -> 16, G -> 24; T (8) -> 0 by the sign selector

// LDS-DMA of 16 bytes per lane: global address g + OFF, LDS address m0 + OFF + 16 * lane (issued where the compiler cannot see it)
template <int OFF>
__device__ __forceinline__ void dma16_off(const void *g, uint32_t m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(g), "s"(m0v), "n"(OFF) : "memory", "m0");
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
// counter word at wb + 4 * B  +=  1 << (byte J of sh)
template <int J, int B>
__device__ __forceinline__ void f6_add(uint32_t wb, uint32_t sh, uint32_t one) {
    uint32_t t;
    if (J == 0) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(one), "v"(wb), "n"(4 * B) : "memory");
    if (J == 1) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(one), "v"(wb), "n"(4 * B) : "memory");
    if (J == 2) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(one), "v"(wb), "n"(4 * B) : "memory");
    if (J == 3) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\tds_add_u32 %3, %0 offset:%4" : "=&v"(t) : "v"(sh), "v"(one), "v"(wb), "n"(4 * B) : "memory");
}
// the 16 bases of a piece (masked codes m) into the counter words from wb on
__device__ __forceinline__ void f6_count16(const uint2 &m, uint32_t wb, uint32_t one, uint32_t &badacc) {
    const uint32_t se0 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, (m.x >> 4) & 0x0F0F0F0Fu);      // even bases of the first half
    const uint32_t so0 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, m.x & 0x0F0F0F0Fu);
    const uint32_t se1 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, (m.y >> 4) & 0x0F0F0F0Fu);
    const uint32_t so1 = __builtin_amdgcn_perm(F6_LUT_HI, F6_LUT_LO, m.y & 0x0F0F0F0Fu);
    // codes that are not one of A C G T: bit 7 of the shift count, or code 12 (which the permute turns into a zero like T's)
    badacc |= ((se0 | so0 | se1 | so1) & 0x80808080u) | (((m.x & (m.x >> 1)) | (m.y & (m.y >> 1))) & 0x44444444u);
    f6_add<0, 0>(wb, se0, one);  f6_add<0, 1>(wb, so0, one);  f6_add<1, 2>(wb, se0, one);  f6_add<1, 3>(wb, so0, one);
    f6_add<2, 4>(wb, se0, one);  f6_add<2, 5>(wb, so0, one);  f6_add<3, 6>(wb, se0, one);  f6_add<3, 7>(wb, so0, one);
    f6_add<0, 8>(wb, se1, one);  f6_add<0, 9>(wb, so1, one);  f6_add<1, 10>(wb, se1, one); f6_add<1, 11>(wb, so1, one);
    f6_add<2, 12>(wb, se1, one); f6_add<2, 13>(wb, so1, one); f6_add<3, 14>(wb, se1, one); f6_add<3, 15>(wb, so1, one);
}

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

template <int W>
__global__ void __launch_bounds__(F6_WAVES * 64, 2)
k_fast6(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *counts, EventBuf eb, uint32_t *glist, uint32_t *gcnt,
        uint32_t *clist, int reads_per_block) {
    // separate LDS objects: the compiler only builds alias scopes per LDS variable
    __shared__ uint4 s_q[F6_WAVES][F6_QB / 16];                       // per wave: the tile's qualities, chunk-major
    __shared__ uint4 s_s[F6_WAVES][F6_SB / 16];                       // per wave: its packed bases, chunk-major; piece-major once masked
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
        constexpr int R = 8;                                       // groups of 64 reads whose loads are in flight together
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
    const uint32_t nTS = (nS + 63u) >> 6, nTI = (nI + 63u) >> 6, n_tb = nTS + nTI;

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
    const uint32_t rep = ((uint32_t)lane >> 2) & (uint32_t)(F6_REP - 1);
    const uint32_t wrep = (uint32_t)(uintptr_t)((lds_u8 *)pwin + rep * (uint32_t)(F6_REPW * 4));
    const uint32_t qb_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u8 *)s_q[wave]);
    const uint32_t sb_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u8 *)s_s[wave]);
    const lds_u8 *const qrow = (const lds_u8 *)s_q[wave] + 1024 * (lane >> 5) + 32 * (lane & 31);      // + 2048 (p >> 1) + 16 (p & 1): piece p of the lane's row
    lds_u8 *const srow16 = (lds_u8 *)s_s[wave] + 16 * lane;          // + 1024 c: raw chunk c (32 bases) of the lane's row
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
    auto load_cig = [&](const F6Hdr &h) {
        Cg c;
        const uint32_t nops = h.nops();
#pragma unroll
        for (uint32_t k = 0; k < 5u; ++k) c.w[k] = rd.cig[h.c0 + (k < nops ? k : 0u)];        // (words past the read's own repeat its first)
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
        dma16_off<0>(qe, qb_w);                dma16_off<0>(qo, qb_w + 1024u);
        dma16_off<32>(qe, qb_w + 2048u - 32u);   dma16_off<32>(qo, qb_w + 3072u - 32u);
        dma16_off<64>(qe, qb_w + 4096u - 64u);   dma16_off<64>(qo, qb_w + 5120u - 64u);
        dma16_off<96>(qe, qb_w + 6144u - 96u);   dma16_off<96>(qo, qb_w + 7168u - 96u);
        dma16_off<128>(qe, qb_w + 8192u - 128u); dma16_off<128>(qo, qb_w + 9216u - 128u);
    };
    // rows of packed bases: instruction c moves chunk c of the lane's own row to 1024 c + 16 lane
    auto issue_s = [&](const F6Hdr &h) {
        const uint8_t *sr = rd.seq + (int64_t)h.o8 * 4;
        if (__ballot((uint64_t)h.o8 + 20u > (uint64_t)q_tot8 + 4u)) {
            const uint32_t lim = (q_tot8 - h.o8) * 4u;
#pragma unroll
            for (int c = 0; c < F6_NP / 2; ++c) dma16_off<0>(sr + ((uint32_t)(16 * c) < lim ? (uint32_t)(16 * c) : lim), sb_w + 1024u * (uint32_t)c);
            return;
        }
        dma16_off<0>(sr, sb_w); dma16_off<16>(sr, sb_w + 1008u); dma16_off<32>(sr, sb_w + 2016u); dma16_off<48>(sr, sb_w + 3024u); dma16_off<64>(sr, sb_w + 4032u);
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
    // results of a tile are stored half a tile later (stores and loads retire through one counter)
    enum : uint32_t { P_STORED = 1u << 24, P_LIST = 1u << 25, P_STATUS_ONLY = 1u << 26 };
    struct Pend { uint32_t i, slot_lo; int32_t pos, reflen; uint32_t meta, cw0, cw1, cw2, cw3, cw4; };
    auto store_pending = [&](const Pend &r) {
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
        const bool has = (r.meta & P_LIST) != 0;
        const unsigned long long m = __ballot(has);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add((lds_u32 *)&s_gcur, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (has) glist[(size_t)rb + base + __popcll(m & ((1ull << lane) - 1ull))] = r.i | ((r.meta & P_STATUS_ONLY) ? GL_STATUS_ONLY : 0u);
        }
    };

    // ---- prologue of the pipeline ------------------------------------------------------------------------------------------
    uint32_t tk0 = take_ticket(), tk1 = take_ticket(), tk2 = take_ticket();
    uint32_t e0 = entry_of(tk0), e1 = entry_of(tk1), e2 = entry_of(tk2);
    F6Hdr h0 = pack_hdr(load_raw(e0), e0);
    Raw r1 = load_raw(e1);
    Cg c0w = load_cig(h0);
    if (tk0 < n_tb) issue_q(h0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
    Shape sh0 = shape_of(h0, c0w, tk0 >= nTS);
    Tabs t0 = load_tabs(h0, sh0);
    if (tk0 < n_tb) issue_s(h0);
    F6Hdr h1 = pack_hdr(r1, e1);
    Raw r2 = load_raw(e2);
    Pend pend{0u, 0u, 0, 0, 0u, 0u, 0u, 0u, 0u, 0u};

    auto turn = [&](auto itag) {
        constexpr bool ITILE = decltype(itag)::value;                   // a tile of indel reads
        const F6Hdr h = h0;
        Shape shp = sh0;
        if (!ITILE) { shp.s.kind = 0; shp.s.k = 0; shp.s.m2 = 0; }      // (constants for the compiler: the closed forms fold)
        const int64_t i = (int64_t)h.idx;
        const int32_t pos = h.pos;
        const uint32_t lseq = h.lseq(), flag = h.flag();
        const bool rev = (flag & 0x10u) != 0;
        // ---- the wave's packed window: fold and re-anchor when the tile has moved on, or before a byte could overflow.
        // The tile's leftmost read anchors it (the order inside a list is almost, not exactly, the batch's)
        int32_t minpos;
        {
            int32_t v = h.valid() ? pos : 0x7FFFFFFF;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int32_t u = __shfl_xor(v, o); v = u < v ? u : v; }
            minpos = __builtin_amdgcn_readfirstlane(v);
            const int32_t want = (minpos < 16 ? 0 : minpos - 16) & ~15;
            if (pw_tiles >= F6_FLUSH || want < pw_base || want - pw_base >= 32) {
                if (pw_tiles) fold();
                set_window(minpos); pw_tiles = 0;
            }
            ++pw_tiles;
        }
        // ---- pass 1a: failing windows of the whole read, pieces in their natural order (the quality rows were waited for at
        // the end of the previous turn) --------------------------------------------------------------------------------------
        uint32_t F[F6_NP / 2];
        uint32_t fb;
        {
            uint4 q;
            { const amp_u32x4 a = *(const lds_u32x4 *)(qrow); q = make_uint4(a.x, a.y, a.z, a.w); }
            fb = q.x & 0xFFu;
#pragma unroll
            for (int k = 0; k < F6_NP; ++k) {
                uint4 nq = q;
                if (k + 1 < F6_NP) { const amp_u32x4 a = *(const lds_u32x4 *)(qrow + 2048 * ((k + 1) >> 1) + 16 * ((k + 1) & 1)); nq = make_uint4(a.x, a.y, a.z, a.w); }
                const uint32_t f16 = P.do_trim ? f6_fail16<W>(q, nq, thr, nthr) : 0u;
                if (k & 1) F[k >> 1] |= f16 << 16; else F[k >> 1] = f16;
                q = nq;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the table entries and the rows of packed bases requested at the end of the previous turn
        const Tabs tA = t0;
        // ---- primer clips in closed form (A:450-558) ---------------------------------------------------------------------
        Bf s = shp.s;
        const bool shaped = shp.ok;
        const bool in_ref = (uint32_t)pos < G && (uint32_t)(pos + shp.refspan - 1) < G;          // A:450-451
        const int32_t q_ins = s.kind ? s.a + s.m1 : 0;                 // query index of the first inserted base / of the base behind the deletion
        TrimState ts{pos, 1, 0u, 0};
        {
            const bool trim = shaped & (P.do_trim != 0), use = trim & in_ref;
            ts.err = (trim & !in_ref) ? AMP_RS_INDEX_REF : 0;
            int32_t p2 = pos; uint32_t f2 = 0u;
            const Bf sp = bf_trim_primers(s, p2, f2, flag, h.isize_flag(), (int32_t)lseq, tA.L, tA.R);
            s = bf_pick(use, sp, s); ts.pos = use ? p2 : pos; ts.flags = use ? f2 : 0u;
        }
        const bool scan = shaped & (P.do_trim != 0) & (ts.err == 0) & !s.punt;
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
        // ---- pass 1b: the codes of the low-quality bases become zero; the masked pieces go back piece-major -------------------
        uint4 iq16 = make_uint4(0u, 0u, 0u, 0u);
        const int32_t g_ins = q_ins & ~7;
        if (ITILE) iq16 = row16(g_ins);
        uint32_t zacc = 0;                           // a code 0 ('=') under a good quality: bit 3 of some nibble (pass 2 cannot tell it from a masked base)
#pragma unroll
        for (int c = 0; c < F6_NP / 2; ++c) {
            const amp_u32x4 qa = *(const lds_u32x4 *)(qrow + 2048 * c), qb = *(const lds_u32x4 *)(qrow + 2048 * c + 16);
            const amp_u32x4 sq = *(const lds_u32x4 *)(srow16 + 1024 * c);
            const uint32_t k0 = f6_keep8(f6_ok80(qa.x, mqb), f6_ok80(qa.y, mqb)), k1 = f6_keep8(f6_ok80(qa.z, mqb), f6_ok80(qa.w, mqb));
            const uint32_t k2 = f6_keep8(f6_ok80(qb.x, mqb), f6_ok80(qb.y, mqb)), k3 = f6_keep8(f6_ok80(qb.z, mqb), f6_ok80(qb.w, mqb));
            { const uint32_t w0 = sq.x | ~k0, w1 = sq.y | ~k1, w2 = sq.z | ~k2, w3 = sq.w | ~k3;      // has-zero-nibble over the kept codes
              zacc |= ((w0 - 0x11111111u) & ~w0) | ((w1 - 0x11111111u) & ~w1) | ((w2 - 0x11111111u) & ~w2) | ((w3 - 0x11111111u) & ~w3); }
            *(lds_u32x2 *)(srow8 + 1024 * c) = amp_u32x2{sq.x & k0, sq.y & k1};
            *(lds_u32x2 *)(srow8 + 1024 * c + 512) = amp_u32x2{sq.z & k2, sq.w & k3};
        }
        zacc &= 0x88888888u;
        wave_sync();                                 // both buffers have been read: the quality image is free
        // ---- requests, first half: the next tile's CIGAR words and quality rows, the header of the tile behind it, the list
        // entries of the one behind that; and this turn's stores of the previous tile's results ---------------------------------
        store_pending(pend);
        const uint32_t tk3 = take_ticket();
        const Cg c1w = load_cig(h1);
        if (tk1 < n_tb) issue_q(h1);
        const F6Hdr h2 = pack_hdr(r2, e2);
        const uint32_t e3 = entry_of(tk3);
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
            mask_piece(qa1 >> 4, any1);
            mask_piece((qb1 - 1) >> 4, any1);
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
        // ---- pass 2: the masked codes into the wave's packed window.  A tile is counted in passes: every pass takes the lanes
        // whose counted positions lie inside the window, then the window is folded and anchored at the leftmost lane left.
        // Lane l works on piece (k + l) mod np in step k: the lanes of a pile of reads do not add to one address at a time -------
        const uint32_t np = (lseq + 15u) >> 4;
        const uint32_t npc = np < 1u ? 1u : (np > (uint32_t)F6_NP ? (uint32_t)F6_NP : np);
        const uint32_t rot = (uint32_t)lane % npc;
        uint32_t badacc = 0;
        // (slow_all: also a read no window position takes -- the first / last 16 positions of the reference)
        bool todo = counted && !slow_all;
        for (bool first_pass = true;; first_pass = false) {
            const unsigned long long tm = __ballot(todo);
            if (!tm) break;
            // (the leftmost of the lanes that are left)
            int32_t v = todo ? ts.pos : 0x7FFFFFFF;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int32_t u = __shfl_xor(v, o); v = u < v ? u : v; }
            const int32_t lead_pos = __builtin_amdgcn_readfirstlane(v);
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
#pragma unroll
            for (int k = 0; k < F6_NP; ++k) {
                uint32_t p = (uint32_t)k + rot;
                p = p >= npc ? p - npc : p;
                const bool slot = (uint32_t)k < npc;
                const int32_t j0 = (int32_t)(p * 16u);
                const amp_u32x2 x = *(const lds_u32x2 *)(srow8 + 512 * (slot ? p : 0u));
                const bool second = ITILE && (j0 >= b1 || b1 <= a1);      // a piece behind the first range belongs to the second
                const bool in = slot && (second ? (j0 < b2 && j0 + 16 > a2 && a2 < b2) : (j0 < b1 && j0 + 16 > a1 && a1 < b1));
                int32_t d0 = (second ? dbase2 : dbase1) + j0;
                d0 = d0 < 0 ? 0 : (d0 > F6_PW - 16 ? F6_PW - 16 : d0);
                const uint2 m = in ? make_uint2(x.x, x.y) : make_uint2(0u, 0u);
                f6_count16(m, wrep + (uint32_t)d0 * 4u, one, badacc);
            }
            todo = todo && !now && !slow_all;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_sync();                                 // the base image has been read: free
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
        // ---- results and hand-over to the general pass: kept for the next turn ---------------------------------------------------
        {
            uint32_t meta = ncig | ((uint32_t)ts.err << 8) | ((ts.err ? 0u : ts.flags) << 16) | (stored ? P_STORED : 0u);
            if (general) meta |= P_LIST;
            else if (stored && !ts.err && P.do_count && want_status) meta |= P_LIST | P_STATUS_ONLY;      // a base could not be counted: exact status wanted
            pend = Pend{(uint32_t)i, h.c0, ts.pos, reflen, meta, cw[0], cw[1], cw[2], cw[3], cw[4]};
        }
        // ---- requests, second half: everything asked for behind pass 1b has arrived; the next tile's table entries and rows of
        // packed bases ------------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        tk0 = tk1; tk1 = tk2; tk2 = tk3;
        h0 = h1; h1 = h2;
        e2 = e3;
        r2 = load_raw(e3);
        sh0 = shape_of(h0, c1w, tk0 >= nTS);
        t0 = load_tabs(h0, sh0);
        if (tk0 < n_tb) issue_s(h0);
    };
    while (tk0 < n_tb) {
        if (tk0 >= nTS) turn(std::true_type{}); else turn(std::false_type{});
    }
    store_pending(pend);
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
    if (tid == 0) gcnt[blockIdx.x] = s_gcur;
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
    switch (P.window) {
        case 1: k_fast6<1><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        case 2: k_fast6<2><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        case 3: k_fast6<3><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        case 4: k_fast6<4><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        case 5: k_fast6<5><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        case 6: k_fast6<6><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        case 7: k_fast6<7><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
        default: k_fast6<8><<<g, t, 0, stream>>>(P, rd, read_base, out, counts, eb, glist, gcnt, clist, rpb); break;
    }
    return (int)hipGetLastError();
}

}  // namespace amp
