// amp_fast5.hpp -- the fast kernel, second generation (variant 5, the default): trim + pileup of the reads whose CIGAR has
// the shape [S a][M m1]([I|D k][M m2])[S c], one lane per read, every byte of the batch loaded once.  CDNA4 / gfx950.
//
// Same decomposition as amp_fast.hpp (variant 4: a wave takes a TILE of 64 consecutive reads of the sorted batch, lane =
// read, closed-form trims from amp_read.hpp, packed per-wave counter windows in LDS) with a leaner instruction stream.
// What the counters of variant 4 showed (profiles/r03_*): the kernel is bound by instruction issue -- a wave issues one
// instruction per ~4.4 cycles, a tile cost ~4,300 of them, two waves per SIMD slow each other by a quarter in EVERY
// phase -- not by HBM, LDS or the I-cache.  So this version spends instructions, not bytes:
//   * a tile's qualities AND packed bases arrive by LDS-DMA in two staging buffers and are consumed from there, 16 bases
//     (a PIECE) at a time: no register copy of the read (60 VGPRs in variant 4), no register-staged base bytes, no LDS
//     write + read-back of the bases
//   * one pass over the qualities turns every piece into 16 failing-window bits + 16 "quality >= min_quality" bits
//     (one VGPR per piece); the staging buffer is then free for the next tile's qualities while the bases are counted
//   * the counted-base test, the range mask and the counter shift of a piece are computed from those bits with byte
//     tricks that need no per-base work besides the SDWA shift + ds_add pair; codes outside A C G T are found with one
//     popcount / has-zero-nibble test per piece (pad nibbles of the staged rows are patched to a valid code first)
//   * every vector-memory operation of the loop that returns data is issued by inline assembly or as a plain load that
//     is first touched behind the ONE wait at the top of the loop, so the compiler never drains the prefetches early
//     (its wait-count pass answers a touched in-flight register with vmcnt(0)): the loop has two full drains, both of
//     loads issued a phase or more before (top: next qualities, primer-table entries, CIGAR words, header;
//     before counting: this tile's bases)
//   * results are stored in the tile they belong to (no one-tile deferral, no packed `pending' registers)
// Reads it does not take (other CIGARs, QUAL '*', more than F5_MAXLEN bases, rows that do not fit the staging buffer)
// go on the block's list for the general pass, exactly as in variant 4.
#pragma once

#include "amp_fast.hpp"
#include "amp_bf.hpp"

namespace amp {

#ifndef AMP_F5_ADD64
#define AMP_F5_ADD64 1      // 64-bit counter adds where a build's replica count allows them (the eight-wave build)
#endif
#ifndef AMP_F5_ABL
#define AMP_F5_ABL 0      // development builds: parts of the kernel switched off to count the rest's instructions (results are wrong on purpose)
#endif
constexpr int F5_NP = 20;                 // 16-base pieces of the longest read taken
constexpr int F5_MAXLEN = 304;            // F5_NP pieces cover it from 8 bases before its start
constexpr int F5_PAD = 16;                // bytes in front of / behind a staged run
// Three builds of the kernel (template parameters WAVES, QRUN, REP; one block per CU, all of its LDS):
//   8 waves, runs of 9,728 quality bytes (64 reads of up to 152 padded bases: every tile of a 150 bp run fits), packed
//     windows of 256 positions in 4 replicas
//   6 waves, runs of 13,312 bytes (64 reads of 208 padded bases on average), windows of 400 positions (a read of 300 bases
//     does not fit 256) in 3 replicas
//   4 waves, runs of 19,456 bytes (64 reads of 304 padded bases: every tile of reads the kernel takes fits), 400 positions
//     in 5 replicas
// The host picks by the batch's mean padded read length; reads of a tile that do not fit its run go to the general pass.
// A counter byte gets at most 4 * ceil(16 / REP) increments per tile (a read covers a position once; lanes go to replicas in
// groups of four): the window is folded every 255 / that many tiles at the latest.
// (Four replicas instead of variant 4's eight: two-way bank conflicts cost nothing, the LDS takes four cycles to receive an
// atomic's operands anyway.)

// LDS-DMA of 16 bytes per lane, issued where the compiler cannot see it (see the head of the file)
__device__ __forceinline__ void dma16(const void *g, const lds_u8 *l) {
    const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)l);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(la) : "memory", "m0");
}

// OR over the lanes of the wave (DPP row shifts and row broadcasts; the result is uniform)
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) {
    uint32_t t = x;
    t |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xF, 0xF, true);      // row_shr:1
    t |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xF, 0xF, true);      // row_shr:2
    t |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xF, true);      // row_shr:4
    t |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xF, 0xF, true);      // row_shr:8
    t |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
    t |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
}

template <int W, int F5_WAVES, int F5_QRUN, int F5_REP, int F5_PW>
__global__ void __launch_bounds__(F5_WAVES * 64, 2)
k_fast5(F_ARGS) {      // (individual arguments, like k_fast: see the note at F_ARGS)
    const KParams P{a_min_quality, a_window, a_do_trim, a_do_count, a_ref_len, a_max_primer_len, a_min_start, a_max_end, (uint32_t)a_epoch};
    const amp_dev_reads rd{a_n_reads, a_pos, a_flag, a_tlen, a_lseq, a_cig_off32, a_cig, a_seq_off8, a_seq, a_qual, 0, 0};
    const DevOut out{a_new_pos, a_new_ncig, a_new_cig, a_o_ref_len, a_trim_flags, a_status};
    const EventBuf eb{a_ev, a_ctr, a_ins_at, a_ev_cap};
    constexpr int F5_REPW = F5_PW + 1;        // words of a replica of the wave's packed window (replica r is skewed by r banks)
    // 64-bit adds (count_piece5q, amp_fast.hpp): a replica is two arrays, for pieces that start on an even / odd window offset; needs an
    // even number of arrays and an odd number of words in each
    constexpr bool F5_ADD64 = AMP_F5_ADD64 && F5_REP % 2 == 0 && F5_REPW % 2 == 1;
    constexpr int F5_NREP = F5_ADD64 ? F5_REP / 2 : F5_REP;               // replicas a lane can be sent to
    constexpr int F5_FLUSH = 255 / (((16 + F5_NREP - 1) / F5_NREP) * 4);      // (lanes go to replicas in groups of four: so many of them add into one array at most)
    constexpr int F5_QB = F5_PAD + F5_QRUN + 2 * F5_PAD, F5_SB = F5_PAD + F5_QRUN / 2 + F5_PAD;      // (a row's last piece is read with the 8 bytes behind it: up to 23 bytes past the run)
    __shared__ uint4 s_q[F5_WAVES][F5_QB / 16];                       // per wave: the tile's quality bytes
    __shared__ uint4 s_s[F5_WAVES][F5_SB / 16];                       // per wave: the tile's packed bases
    __shared__ __attribute__((aligned(8))) uint32_t s_pwin[F5_WAVES][F5_REP * F5_REPW];           // per wave: packed counters, byte c of a word = base c (A C G T)
    __shared__ uint32_t s_bwin[F_BPL * F_BW];                         // the block's window, 32-bit counters
    __shared__ uint32_t s_ticket, s_gcur;
    unsigned long long *const ctr = eb.ctr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t n = rd.n_reads;
    const int64_t rb = (int64_t)blockIdx.x * reads_per_block;
    const int64_t re = rb + reads_per_block < n ? rb + reads_per_block : n;
    lds_u32 *const bwin = (lds_u32 *)s_bwin;
    lds_u32 *const pwin = (lds_u32 *)s_pwin[wave];
    for (int i = tid; i < F_BPL * F_BW; i += F5_WAVES * 64) bwin[i] = 0;
    for (int i = lane; i < F5_REP * F5_REPW; i += 64) pwin[i] = 0;
    // every nibble of the base staging buffer starts as a valid code (the bytes in front of a run are read by the
    // lanes whose pieces start 8 bases early, and are never written again)
    for (int i = lane; i < F5_SB / 4; i += 64) ((lds_u32 *)s_s[wave])[i] = 0x11111111u;
    if (tid == 0) { s_ticket = 0; s_gcur = 0; }
    if (tid == 0 && blockIdx.x == 0) { eb.ctr[26] = 0ull; eb.ctr[27] = 0ull; eb.ctr[28] = 0ull; }      // k_gcompact / k_long's counters (amp_wave.hpp)
    int32_t bw_base = rb < n ? rd.pos[rb] : 0;
    bw_base = (bw_base < 16 ? 0 : bw_base - 16) & ~15;
    __syncthreads();

    const int32_t mq = P.min_quality;
    const uint32_t mqc = (uint32_t)(mq > 256 ? 256 : mq);
    const uint32_t thr = mqc * (uint32_t)W;
    const uint32_t mqb = (uint32_t)mq * 0x01010101u;             // mq <= 128 (the host sends other runs to the general kernel)
    const uint32_t G = (uint32_t)P.ref_len;
    const uint32_t n_tb = re > rb ? (uint32_t)((re - rb + 63) / 64) : 0u;
    auto take_ticket = [&]() {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((lds_u32 *)&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    uint32_t n_err = 0;
    // bank plan: lane l works on piece (k + l) mod np in step k, starts its pieces 8 bases early when bit 1 of l is
    // set and adds into replica (l >> 2) & 3 (replica r is skewed by r banks)
    const uint32_t rep = (F5_REP & (F5_REP - 1)) ? ((uint32_t)lane >> 2) % (uint32_t)F5_REP : ((uint32_t)lane >> 2) & (uint32_t)(F5_REP - 1);
    const uint32_t phi_lane = ((uint32_t)lane >> 1) & 1u ? 8u : 0u;
    const uint32_t wrep = (uint32_t)(uintptr_t)((lds_u8 *)pwin + (F5_ADD64 ? (rep % (uint32_t)F5_NREP) * (uint32_t)(2 * F5_REPW * 4) : rep * (uint32_t)(F5_REPW * 4)));
    auto count5 = [&](const uint2 &sq_, uint32_t m_, int32_t d0_, int32_t lim_) -> uint32_t {
        if constexpr (F5_ADD64) return count_piece5q(sq_, m_, d0_, lim_, wrep, (uint32_t)(F5_REPW * 4));
        else return count_piece5(sq_, m_, d0_, lim_, wrep);
    };
    lds_u8 *const qst = (lds_u8 *)s_q[wave] + F5_PAD;
    lds_u8 *const sst = (lds_u8 *)s_s[wave] + F5_PAD;
    int32_t pw_base = 0;
    int pw_tiles = F5_FLUSH;
    const unsigned ev_shard = blockIdx.x & (EV_SHARDS - 1);
    amp_ins_event *const ev_list = eb.ev + (size_t)ev_shard * (size_t)eb.cap;
    unsigned long long ev_base = 0;
    uint32_t ev_left = 0;

    // folds the wave's packed window into the block's 32-bit window (or the global table) and clears it
    auto fold = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the adds of the counting phase are invisible to the compiler's wait counts)
        wave_sync();
#pragma unroll 1
        for (int idx = lane; idx < F5_PW; idx += 64) {
            uint32_t ag = 0, ct = 0;                                 // A | G << 16, C | T << 16
#pragma unroll
            for (int r = 0; r < F5_REP; ++r) {
                const uint32_t w = pwin[r * F5_REPW + idx];
                pwin[r * F5_REPW + idx] = 0;
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
    auto pad_events = [&]() {
        if ((uint32_t)lane < ev_left && (long long)(ev_base + (unsigned)lane) < eb.cap) ev_list[ev_base + (unsigned)lane] = amp_ins_event{-1, 0u, 0, 0};
    };

    struct Hdr { int32_t pos, tlen; uint32_t lseq, flag, c0, c1, o8; };      // as loaded
    // ... and as kept: lf = l_seq (saturated at 0xFFFF) | paired << 16 | reverse << 17 | the template-length test of A:452 << 18 |
    // number of CIGAR ops (saturated at 7) << 19 | lane holds a read of the block << 22
    struct HdrP { int32_t pos; uint32_t lf, c0, o8;
        __device__ uint32_t lseq() const { return lf & 0xFFFFu; }
        __device__ uint32_t nops() const { return (lf >> 19) & 7u; }
        __device__ uint32_t flag() const { return ((lf >> 16) & 1u) | (((lf >> 17) & 1u) << 4); }
        __device__ bool isize_flag() const { return (lf >> 18) & 1u; }
        __device__ bool valid() const { return (lf >> 22) & 1u; } };
    auto load_hdr = [&](int64_t t0) {
        Hdr h{0, 0, 0u, 0u, 0u, 0u, 0u};
        const int64_t i = t0 + lane;
        if (i < re) {
            h.pos = rd.pos[i]; h.flag = rd.flag[i]; h.tlen = rd.tlen[i]; h.lseq = rd.lseq[i];
            h.c0 = rd.cig_off32[i]; h.c1 = rd.cig_off32[i + 1]; h.o8 = rd.seq_off8[i];
        }
        return h;
    };
    auto pack_hdr = [&](const Hdr &h, int64_t t0) {
        const uint32_t nn = h.c1 - h.c0, at = (uint32_t)(h.tlen < 0 ? -(int64_t)h.tlen : (int64_t)h.tlen);
        const bool isz = ((int64_t)at - P.max_primer_len) > (int64_t)h.lseq;                                  // A:452
        return HdrP{h.pos, (h.lseq > 0xFFFFu ? 0xFFFFu : h.lseq) | ((h.flag & 1u) << 16) | (((h.flag >> 4) & 1u) << 17) | ((isz ? 1u : 0u) << 18) |
                               ((nn > 7u ? 7u : nn) << 19) | ((t0 + lane < re ? 1u : 0u) << 22), h.c0, h.o8};
    };
    struct Cg { uint32_t w[5]; };
    auto load_cig = [&](const HdrP &h) {
        Cg c{{0u, 0u, 0u, 0u, 0u}};
        const uint32_t nops = h.nops();
        if (nops >= 1u && nops <= 5u) {
#pragma unroll
            for (uint32_t k = 0; k < 5u; ++k) c.w[k] = rd.cig[h.c0 + (k < nops ? k : 0u)];
        }
        return c;
    };
    // the tile's run: the bytes of its leading reads that fit the staging buffer (64 reads of up to 152 bases always do)
    struct Geo { uint32_t np, row, Tq, m0; int ntake; bool solo, fastq; };
    auto geometry = [&](const HdrP &h) {
        Geo g;
        const bool valid = h.valid();
        const uint32_t hl = h.lseq();
        const bool shortq = valid && hl >= 1u && hl <= (uint32_t)F5_MAXLEN;
        g.np = shortq ? (hl + phi_lane + 15u) >> 4 : 1u;
        g.m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.o8);
        g.row = (h.o8 - g.m0) * 8u;
        const uint32_t nch = (hl + 7u) >> 3;
        // a row is read as pieces of 16 bytes plus the 8 bytes behind them: up to 24 bytes past the read's own padded bytes
        const bool fits = valid && h.o8 >= g.m0 && (h.o8 - g.m0) <= (uint32_t)(F5_QRUN / 8) && g.row + 8u * nch <= (uint32_t)F5_QRUN;
        const unsigned long long fitmask = __ballot(fits);
        g.ntake = fitmask == ~0ull ? 64 : __builtin_ctzll(~fitmask);
        g.solo = g.ntake == 0;
        if (g.solo) g.ntake = 1;
        g.Tq = g.solo ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)(g.row + 8u * nch), g.ntake - 1);
        g.fastq = lane < g.ntake && !g.solo && shortq;
        if (!g.fastq) { g.row = 0u; g.np = 1u; }
        return g;
    };
    struct Shape { Bf s; bool ok; int32_t refspan; };
    auto shape_of = [&](const HdrP &h, const Cg &c, bool fastq) {
        Shape r;
        bool ok;
        r.s = bf_from_words5((int)h.nops(), c.w[0], c.w[1], c.w[2], c.w[3], c.w[4], (int32_t)h.lseq(), F_MAXINS, F_MAXDEL, ok);
        r.ok = ok & fastq;
        r.refspan = r.ok ? r.s.m1 + r.s.m2 + r.s.kD() : 1;
        return r;
    };
    struct Tabs { int32_t L, R; };
    auto load_tabs = [&](const HdrP &h, const Shape &sh) {
        Tabs t{-1, -1};
        const bool in_ref = ((uint32_t)h.pos < G) & ((uint32_t)(h.pos + sh.refspan - 1) < G);
        if (sh.ok & (P.do_trim != 0) & in_ref) { t.L = P.max_end[h.pos]; t.R = P.min_start[h.pos + sh.refspan - 1]; }
        return t;
    };
    // LDS-DMA of a run: lane l moves bytes [1024 s + 16 l, + 16) to the same offset of the staging buffer; lanes past the
    // run re-read its end, lanes past the buffer do nothing
    auto issue_run = [&](const uint8_t *run, uint32_t nbytes, lds_u8 *stage, int cap) {
        const uint32_t last = nbytes ? (nbytes - 1u) & ~15u : 0u;
#pragma unroll
        for (int sl = 0; sl < (cap + 1023) / 1024; ++sl) {
            uint32_t off = (uint32_t)(sl * 1024 + lane * 16);
            if (sl * 1024 + 1024 <= cap || (int)off < cap) dma16(run + (off < last ? off : last), stage + sl * 1024);
        }
    };

    uint32_t pw_lim = 0;

    if (AMP_F_STAGGER > 0) {       // (wave w starts w steps late, as in k_fast)
        const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)AMP_F_STAGGER * (unsigned)wave;
        while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(8);
    }
    // ---- prologue: three tiles' headers; CIGAR words of the first two; the first tile's qualities and table entries -----------
    uint32_t tk0 = take_ticket(), tk1 = take_ticket(), tk2 = take_ticket();
    int64_t i0 = rb + 64 * (int64_t)tk0, i1 = rb + 64 * (int64_t)tk1, i2 = rb + 64 * (int64_t)tk2;
    HdrP h0, h1;
    Hdr hR;
    {
        const Hdr a = load_hdr(i0), b = load_hdr(i1);
        hR = load_hdr(i2);
        h0 = pack_hdr(a, i0); h1 = pack_hdr(b, i1);
    }
    Cg cN = load_cig(h1);
    Geo g0 = geometry(h0);
    Shape sh0;
    {
        const Cg c = load_cig(h0);
        issue_run(rd.qual + (int64_t)g0.m0 * 8, g0.Tq, qst, F5_QRUN);
        sh0 = shape_of(h0, c, g0.fastq);
    }
    Tabs tb0 = load_tabs(h0, sh0);
    uint32_t pfA = 0u, pfB = 0u;                   // (L2 prefetch of the next tile's bases: see below)
    while (tk0 < n_tb) {
        // ---- everything issued a phase or more ago has arrived: this tile's qualities and table entries, the next tile's
        // CIGAR words, the header of the tile behind it -------------------------------------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : : "v"(pfA), "v"(pfB));
        // (every register a load of the last turn -- or of the prologue -- wrote is touched HERE, where nothing is in flight: left to its
        // own devices the compiler waits at the first use, behind the issue of this tile's bases, and the wave sits out their latency)
        asm volatile("" : : "v"(tb0.L), "v"(tb0.R), "v"(cN.w[0]), "v"(cN.w[1]), "v"(cN.w[2]), "v"(cN.w[3]), "v"(cN.w[4]));
        asm volatile("" : : "v"(hR.pos), "v"(hR.tlen), "v"(hR.lseq), "v"(hR.flag), "v"(hR.c0), "v"(hR.c1), "v"(hR.o8));
        const HdrP h = h0;
        const Geo g = g0;
        const Shape shp = sh0;
        const Tabs tA = tb0;
        const Geo g1 = geometry(h1);
        const Shape sh1 = shape_of(h1, cN, g1.fastq);
        const HdrP h2 = pack_hdr(hR, i2);
        // this tile's packed bases start moving (their buffer was in use until the end of the last tile)
        issue_run(rd.seq + (int64_t)g.m0 * 4, g.Tq >> 1, sst, F5_QRUN / 2);
        const int64_t i = i0 + lane;
        const int32_t pos = h.pos;
        const uint32_t lseq = h.lseq(), flag = h.flag(), c0 = h.c0, o8 = h.o8;
        const uint32_t np = g.np, phi = g.fastq ? phi_lane : 0u;
        const bool fastq = g.fastq;
        // ---- the wave's packed window: fold and re-anchor when the tile has moved on, or before a byte could overflow
        {
            const int32_t first_pos = __builtin_amdgcn_readfirstlane(pos);
            const int32_t want = (first_pos < 16 ? 0 : first_pos - 16) & ~15;
            if (pw_tiles >= F5_FLUSH || want < pw_base || want - pw_base >= 64) {
                if (pw_tiles) fold();
                pw_base = want; pw_tiles = 0;
            }
            ++pw_tiles;
        }
        pw_lim = (int64_t)G - pw_base >= (int64_t)F5_PW ? (uint32_t)F5_PW : (uint32_t)(G > (uint32_t)pw_base ? G - (uint32_t)pw_base : 0u);
        // ---- primer clips in closed form (A:450-558), branch-free (amp_bf.hpp) --------------------------------------
        const bool shaped = shp.ok;
        const bool in_ref = ((uint32_t)pos < G) & ((uint32_t)(pos + shp.refspan - 1) < G);      // A:450-451
        const bool rev = (flag & 0x10u) != 0;
        const bool trim = shaped & (P.do_trim != 0);
        const int terr = (trim & !in_ref) ? AMP_RS_INDEX_REF : 0;
        const bool use = trim & in_ref;
        int32_t tpos = pos;
        uint32_t tflags = 0u;
        Bf s = shp.s;
        {
            int32_t p2 = pos; uint32_t f2 = 0u;
            const Bf sp = bf_trim_primers(s, p2, f2, flag, h.isize_flag(), (int32_t)lseq, tA.L, tA.R);
            if (!(AMP_F5_ABL & 16)) { s = bf_pick(use, sp, s); tpos = use ? p2 : pos; tflags = use ? f2 : 0u; }
        }
        const bool scan = use & !s.punt;
        // aligned-quality window [lo, hi) in PIECE coordinates (query index + phi)
        int32_t lo, qlen;
        bf_quality_window(s, (int32_t)lseq, lo, qlen);
        lo = scan ? lo + (int32_t)phi : 0; qlen = scan ? qlen : 0;
        const int32_t hi = lo + qlen;
        // ---- pass over the qualities: slot k of the lane is piece (k + rot) mod np of its read.  Per piece: 16 failing-window
        // bits (bit b: the W-byte window starting at base b of the piece sums to less than W * min_quality), of which the
        // first / last inside [lo, hi - W] give the first failing window start (forward) / last failing window end
        // (reverse); and 16 good-quality bits, kept as ok[k] for the counting phase ------------------------------------------
        const uint32_t rot = (uint32_t)lane % np;
        const uint32_t live = wave_or_u32((1u << np) - 1u);                    // bit k: some lane of the tile has a piece in slot k
        const int32_t lrow = (int32_t)g.row - (int32_t)phi;                    // >= -8: the pad in front of the run
        const lds_u8 *const lq = qst + g.row;                                  // the read's qualities in the staging buffer
        uint32_t fo[F5_NP];
        int32_t ffmin = 0x7FFFFFFF, lemax = -1;
        // (the 24 bytes of slot k + 1 are asked for before slot k is worked on: a slot's LDS round trip lies under its predecessor's work)
        auto piece_bytes = [&](int k, amp_u32x2 &a, amp_u32x2 &b, amp_u32x2 &c) {
            uint32_t p = (uint32_t)k + rot;
            p = p >= np ? p - np : p;
            const lds_u8 *src = qst + lrow + (int32_t)(((uint32_t)k < np ? p : np - 1u) * 16u);
            a = *(const lds_u32x2 *)src; b = *(const lds_u32x2 *)(src + 8); c = *(const lds_u32x2 *)(src + 16);
        };
        amp_u32x2 an, bn, cn;
        piece_bytes(0, an, bn, cn);
#pragma unroll
        for (int k = 0; k < F5_NP; ++k) {
            fo[k] = 0u;
            if (!((live >> k) & 1u) || (AMP_F5_ABL & 1)) continue;             // (uniform)
            const amp_u32x2 a = an, b = bn, c = cn;
            if (k + 1 < F5_NP) piece_bytes(k + 1, an, bn, cn);
            uint32_t p = (uint32_t)k + rot;
            p = p >= np ? p - np : p;
            p = (uint32_t)k < np ? p : np;
            const uint4 q = make_uint4(a.x, a.y, b.x, b.y);
            fo[k] = ok_bits16(q, mqb);
            if (P.do_trim) {
                const int32_t j0 = (int32_t)(p * 16u);
                const uint32_t fail = piece_fail_bits<W>(q, make_uint2(c.x, c.y), thr) & range_bits16(lo - j0, hi - W - j0 + 1);      // window starts j0+b must lie in [lo, hi - W]
                const int32_t f1 = j0 + (__builtin_ffs((int)fail) - 1), e1 = j0 + (31 - __builtin_clz(fail)) + W;
                ffmin = ((fail != 0u) & (f1 < ffmin)) ? f1 : ffmin;
                lemax = ((fail != 0u) & (e1 > lemax)) ? e1 : lemax;
            }
        }
        const uint32_t fb = lq[0];                                             // 0xFF = QUAL '*'
        // ---- quality clip (A:589-686) --------------------------------------------------------------------------------
        int32_t iq = rev ? 0 : qlen;
        {
            // the shrinking windows at the 3' end decide when no full window failed (A:575-576, A:637-638)
            int32_t acc = 0;
            const int32_t kmax = qlen < W - 1 ? qlen : W - 1;
            const int32_t qlo = lo - (int32_t)phi, qhi = hi - (int32_t)phi;            // query indices
#pragma unroll
            for (int32_t k = 1; k <= W - 1; ++k) {
                const bool on = k <= kmax;
                const int32_t idx = on ? (rev ? qlo + k - 1 : qhi - k) : 0;
                acc += on ? (int32_t)lq[idx] : 0;
                iq = (on & ((int64_t)acc < (int64_t)mq * k)) ? (rev ? k : qlen - k) : iq;
            }
            iq = (!rev & (ffmin != 0x7FFFFFFF)) ? ffmin - lo : iq;
            iq = (rev & (lemax >= 0)) ? lemax - lo : iq;
        }
        {
            uint32_t f2 = tflags;
            const Bf sq_ = bf_trim_quality(s, tpos, f2, rev, iq, qlen);
            if (!(AMP_F5_ABL & 16)) { s = bf_pick(scan, sq_, s); tflags = scan ? f2 : tflags; }
        }
        // ---- results ---------------------------------------------------------------------------------------------------
        const bool nogo = ((fb & 0xFFu) == 0xFFu) | (s.punt != 0u);               // QUAL '*': the generic code reports it (A:561-562, A:718); a shape the closed forms leave
        const bool general = h.valid() & (!shaped | nogo);                       // (a lane whose bytes did not fit the run included)
        const bool stored = shaped & !nogo;
        const bool okres = stored & (terr == 0);
        const int32_t reflen = okres ? s.ref_len() : 0;
        n_err += (stored & (terr != 0)) ? 1u : 0u;
        const bool counted = okres & (P.do_count != 0);
        // ---- what counting needs of the qualities besides the bits: the inserted bases' (A:730-748), and the good bits of
        // GROUP B = the 16 bases from the 8-aligned start of the second segment, for the piece that holds bases of both
        // segments of an indel read (its second part is counted on its own) --------------------------------------------
        const bool two = counted & (s.kind != 0);
        const int32_t q_seg2 = s.a + s.m1 + s.kI();
        const int32_t g_b = two ? q_seg2 & ~7 : 0;
        uint32_t good = 0, okB = 0;
        if (__ballot(two)) {
            if (two & (s.kind == 1)) {
                uint32_t m = 0;
                for (int32_t j = 0; j < s.k; ++j) m |= ((int32_t)lq[s.a + s.m1 + j] >= mq ? 1u : 0u) << j;      // (a clip may have taken the first inserted bases)
                good = m;
            }
            const amp_u32x2 a = *(const lds_u32x2 *)(lq + g_b), b = *(const lds_u32x2 *)(lq + g_b + 8);
            okB = ok_bits16(make_uint4(a.x, a.y, b.x, b.y), mqb);
        }
        // ---- the tile's bases have arrived; the quality buffer is free: results out, the next tile's loads go out ------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            // [S a][op m1][I|D k][op m2][S c], absent parts left out
            const uint32_t part[5] = {((uint32_t)s.a << 4) | OP_S, ((uint32_t)s.m1 << 4) | s.op, ((uint32_t)s.k << 4) | (s.kind == 1 ? OP_I : OP_D),
                                      ((uint32_t)s.m2 << 4) | s.op, ((uint32_t)s.c << 4) | OP_S};
            const bool has[5] = {bool(okres & (s.a > 0)), bool(okres & (s.m1 > 0)), bool(okres & (s.kind != 0)), bool(okres & (s.kind != 0) & (s.m2 > 0)), bool(okres & (s.c > 0))};
            uint32_t *home = out.new_cig + ((size_t)c0 + 3 * (size_t)i);
            uint32_t nc = 0;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (has[t]) home[nc] = part[t];
                nc += has[t] ? 1u : 0u;
            }
            if (stored) {
                if (out.new_pos) out.new_pos[i] = tpos;
                if (out.new_ncig) out.new_ncig[i] = nc;
                if (out.ref_len) out.ref_len[i] = reflen;
                if (out.trim_flags) out.trim_flags[i] = (uint8_t)(terr ? 0u : tflags);
                if (out.status) out.status[i] = (uint8_t)terr;
            }
        }
        const uint32_t tk3 = take_ticket();
        const int64_t i3 = rb + 64 * (int64_t)tk3;
        issue_run(rd.qual + (int64_t)g1.m0 * 8, g1.Tq, qst, F5_QRUN);
        tb0 = load_tabs(h1, sh1);
        cN = load_cig(h2);
        hR = load_hdr(i3);
        {
            // the next tile's packed bases are pulled into L2 now (one dword of every 128-byte line; plain loads whose values
            // are only "used" behind the wait at the top of the loop): their LDS-DMA can only be issued when this tile's bases
            // have been counted, a third of a tile before they are needed
            const uint32_t nb = g1.Tq >> 1;
            const uint32_t off = (uint32_t)lane * 128u;
            const uint8_t *sb = rd.seq + (int64_t)g1.m0 * 4;
            pfA = *(const uint32_t *)(sb + (off < nb ? off : 0u));
            if (F5_QRUN / 2 > 8192) pfB = *(const uint32_t *)(sb + (off + 8192u < nb ? off + 8192u : 0u));
        }
        // pad nibbles of the staged rows (a row is padded to 8 bases) become a valid code: the test for codes outside
        // A C G T looks at whole pieces
        {
            const uint32_t e = fastq ? lseq & 7u : 0u;                        // bases of the row's last group of 8 (0: the group is full)
            if (e) {
                lds_u32 *w = (lds_u32 *)(sst + (g.row >> 1) + 4u * (lseq >> 3));
                // nibble i of the group sits in byte i >> 1, high nibble first
                const uint32_t x = *w, xs = ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);      // nibble i at bit 4 i
                const uint32_t keep = (1u << (4u * e)) - 1u;
                const uint32_t ys = (xs & keep) | (0x11111111u & ~keep);
                *w = ((ys & 0x0F0F0F0Fu) << 4) | ((ys >> 4) & 0x0F0F0F0Fu);
            }
        }
        wave_sync();

        // ---- counting (A:709-753): the counted query ranges [qa1, qb1) and [qa2, qb2) in piece coordinates, the window
        // offset of piece coordinate 0 for each of them ------------------------------------------------------------------
        const int32_t qa1 = counted ? s.a + (int32_t)phi : 0, qb1 = counted ? qa1 + s.m1 : 0;
        const int32_t qa2 = two ? qb1 + s.kI() : qb1, qb2 = two ? qa2 + s.m2 : qa2;
        const int32_t pos2 = tpos + s.m1 + s.kD();                                 // reference position of the second segment
        bool bad_extra = false;
        // deletion: '-' at each of its positions (A:714-715), through the block's window
        if (two & (s.kind == 2) & !(AMP_F5_ABL & 4)) {
            for (int32_t j = 0; j < s.k; ++j) {
                const int32_t r = tpos + s.m1 + j;
                const uint32_t d = (uint32_t)(r - bw_base);
                if ((uint32_t)r >= G) bad_extra = true;
                else if (d < (uint32_t)F_BW) lds_add_nt(bwin + 4 * F_BW + d, 1u);
                else atomicAdd(&counts[(size_t)r * AMP_NSYM + 5], 1u);
            }
        }
        // insertion (A:730-748): one event per maximal run of good-quality inserted bases
        {
            uint32_t runs = (AMP_F5_ABL & 4) ? 0u : good & ~(good << 1);                              // first base of every run
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
                    const int32_t q0 = s.a + s.m1, r2 = tpos + s.m1, ref_end = tpos + s.m1 + s.m2;
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
        uint32_t redo = 0;                        // pieces (slots) the careful loop has to do; bit F5_NP = group B
        // group B: the part of the second segment that shares a piece with the first
        const int32_t jstar = (qb1 - 1) & ~15;                       // the piece that holds the first segment's last base
        const bool has_b = two & (jstar + 16 > qa2) & (qb2 > qa2);
        const int32_t xbe = qb2 < jstar + 16 ? qb2 : jstar + 16;
        const int32_t jb = g_b + (int32_t)phi;
        // The bases go into the wave's packed window, F_PW positions from pw_base; a tile whose reads lie further apart (the
        // step from one pile of reads to the next) is counted in PASSES: fold, re-anchor at the first lane left.
        const int32_t end_pos = two ? pos2 + s.m2 : tpos + s.m1;                   // one past the last counted position
        const lds_u8 *const lsrow = sst + (int32_t)(g.row >> 1) - (int32_t)(phi >> 1);
        bool todo = counted;
        for (bool first_pass = true;; first_pass = false) {
            const unsigned long long tm = __ballot(todo);
            if (!tm) break;
            const int lead = __builtin_ctzll(tm);
            if (!first_pass) {
                fold();
                const int32_t lead_pos = __builtin_amdgcn_readlane(pos, lead);
                pw_base = (lead_pos < 16 ? 0 : lead_pos - 16) & ~15; pw_tiles = 1;
                pw_lim = (int64_t)G - pw_base >= (int64_t)F5_PW ? (uint32_t)F5_PW : (uint32_t)(G > (uint32_t)pw_base ? G - (uint32_t)pw_base : 0u);
            }
            const int32_t lim16 = (int32_t)pw_lim - 16;
            const bool fits = (tpos - pw_base >= 16) & (end_pos - pw_base + 16 <= (int32_t)pw_lim);
            const bool now = todo & (fits | (lane == lead));
            const int32_t a1 = now ? qa1 : 0, b1 = now ? qb1 : 0, a2 = now ? qa2 : 0, b2 = now ? qb2 : 0;
            const int32_t dbase1 = tpos - pw_base - qa1, dbase2 = pos2 - pw_base - qa2;
            if (__ballot(has_b & now)) {
                const lds_u8 *sp = sst + (g.row >> 1) + (uint32_t)(g_b >> 1);
                const uint2 sq = make_uint2(*(const lds_u32 *)sp, *(const lds_u32 *)(sp + 4));
                const uint32_t mB = (has_b & now) ? okB & range_bits16(qa2 - jb, xbe - jb) : 0u;
                redo |= count5(sq, mB, dbase2 + jb, lim16) << F5_NP;
            }
            // The packed bases of FIVE pieces are read before their adds: LDS operations of a wave complete in order, so a read behind
            // the sixteen adds of a piece waits for every one of them (the compiler cannot see the adds and waits with lgkmcnt(0)); read
            // piece by piece the loop drained the LDS queue once per slot (amp_fast7.hpp, whose lanes have ten pieces at most, reads all)
#pragma unroll
            for (int c = 0; c < F5_NP; c += 5) {
                if (!((live >> c) & 1u) || (AMP_F5_ABL & 2)) continue;      // (uniform; the live slots are the low ones)
                uint2 sqv[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    uint32_t p = (uint32_t)(c + j) + rot;
                    p = p >= np ? p - np : p;
                    const lds_u8 *sp = lsrow + ((uint32_t)(c + j) < np ? p : np - 1u) * 8u;
                    sqv[j] = make_uint2(*(const lds_u32 *)sp, *(const lds_u32 *)(sp + 4));
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int k = c + j;
                    if (!((live >> k) & 1u)) continue;
                    uint32_t p = (uint32_t)k + rot;
                    p = p >= np ? p - np : p;
                    p = (uint32_t)k < np ? p : np;
                    const int32_t j0 = (int32_t)(p * 16u);
                    const bool second = j0 >= qb1;                              // a piece behind the first segment belongs to the second
                    const uint32_t rng = range_bits16((second ? a2 : a1) - j0, (second ? b2 : b1) - j0);
                    redo |= count5(sqv[j], fo[k] & rng, (second ? dbase2 : dbase1) + j0, lim16) << k;
                }
            }
            todo = todo & !now;
        }
        bool want_status = bad_extra;
        if (__ballot(redo != 0u)) {
            // careful loop (rare): bases of the flagged pieces one by one, straight from memory into the 32-bit counters
            if (redo) {
                const uint8_t *qrow = rd.qual + (int64_t)o8 * 8;
                const uint8_t *srow = rd.seq + (int64_t)o8 * 4;
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
                for (int k = 0; k < F5_NP; ++k) {
                    if (!((redo >> k) & 1u)) continue;
                    uint32_t p = (uint32_t)k + rot;
                    p = p >= np ? p - np : p;
                    const int32_t j0 = (int32_t)(p * 16u) - (int32_t)phi;
                    const bool second = j0 + (int32_t)phi >= qb1;
                    const int32_t sa = second ? a2 : a1, sb_ = second ? b2 : b1;
                    careful(j0 < sa ? sa : j0, j0 + 16 < sb_ ? j0 + 16 : sb_, sa, second ? pos2 : tpos);
                }
                if ((redo >> F5_NP) & 1u) careful(a2, xbe - (int32_t)phi, a2, pos2);
            }
        }
        // ---- hand-over to the general pass: the block's segment of the list -----------------------------------------------
        {
            const bool status_only = !general & counted & want_status;        // a base could not be counted: exact status wanted
            const bool has = general | status_only;
            const unsigned long long m = __ballot(has);
            if (m) {
                uint32_t base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add((lds_u32 *)&s_gcur, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (has) glist[(size_t)rb + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i | (status_only ? GL_STATUS_ONLY : 0u);
            }
        }
        // ---- next tile ------------------------------------------------------------------------------------------------
        h0 = h1; h1 = h2; g0 = g1; sh0 = sh1;
        i0 = i1; i1 = i2; i2 = i3; tk0 = tk1; tk1 = tk2; tk2 = tk3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the loads issued for a tile that does not exist)
    pad_events();
    if (pw_tiles && n_tb) fold();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < F_BPL * F_BW; i += F5_WAVES * 64) {
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
    if (n_err) atomicAdd(&ctr[2], (unsigned long long)n_err);
    if (tid == 0) { gcnt[blockIdx.x] = s_gcur; if (s_gcur) eb.ctr[29] = (unsigned long long)P.epoch; }      // (every block writes the same value)
}

struct Fast5Cfg { int waves, qrun; };
// which build: by the mean padded read length of the batch (bases, a multiple of 8 per read)
static inline Fast5Cfg fast5_cfg(int64_t n_reads, int64_t n_bases_padded, int window) {
    const int64_t mean_pad = n_reads > 0 ? (n_bases_padded + n_reads - 1) / n_reads : 0;
    if (mean_pad <= 152 || window != 4) return Fast5Cfg{8, 9728};          // (the other two are built for the default window only)
    if (mean_pad <= 192) return Fast5Cfg{6, 13312};
    return Fast5Cfg{4, 19456};
}
static inline FastGrid fast5_grid(int64_t n_reads, int n_cu, const Fast5Cfg &cf) {
    int64_t rpb = (n_reads + (int64_t)n_cu - 1) / (int64_t)n_cu;
    rpb = ((rpb + 63) / 64) * 64;
    if (rpb < 2 * cf.waves * 64) rpb = 2 * cf.waves * 64;
    return FastGrid{(n_reads + rpb - 1) / rpb, rpb};
}

static inline int fast5_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out, uint32_t *counts,
                               const EventBuf &eb, uint32_t *glist, uint32_t *gcnt, const FastGrid &fg, const Fast5Cfg &cf, hipStream_t stream) {
    const unsigned g = (unsigned)fg.grid;
    const int rpb = (int)fg.rpb;
#define AMP_F5_GO(w, wv, qr, rp, pw) k_fast5<w, wv, qr, rp, pw><<<g, wv * 64, 0, stream>>>(F_ARGS_PASS(P, rd, out, eb, rpb))
    // (replicas / positions of the packed window of the six- and the four-wave build: what the LDS holds.  400 positions take a read of
    //  304 bases + 16 in front + the spread of a tile's starts; a third / fifth replica instead of positions 400..511 is worth 3 %
    //  on 176 and 250 bp runs: the counting adds are bound by LDS bank conflicts, see amp_fast7.hpp)
#ifndef AMP_F5_R6
#define AMP_F5_R6 3
#define AMP_F5_P6 400
#define AMP_F5_R4 5
#define AMP_F5_P4 400
#endif
    if (cf.waves == 6) AMP_F5_GO(4, 6, 13312, AMP_F5_R6, AMP_F5_P6);
    else if (cf.waves == 4) AMP_F5_GO(4, 4, 19456, AMP_F5_R4, AMP_F5_P4);
    else switch (P.window) {
        case 1: AMP_F5_GO(1, 8, 9728, 4, 256); break;
        case 2: AMP_F5_GO(2, 8, 9728, 4, 256); break;
        case 3: AMP_F5_GO(3, 8, 9728, 4, 256); break;
        case 4: AMP_F5_GO(4, 8, 9728, 4, 256); break;
        case 5: AMP_F5_GO(5, 8, 9728, 4, 256); break;
        case 6: AMP_F5_GO(6, 8, 9728, 4, 256); break;
        case 7: AMP_F5_GO(7, 8, 9728, 4, 256); break;
        default: AMP_F5_GO(8, 8, 9728, 4, 256); break;
    }
#undef AMP_F5_GO
    return (int)hipGetLastError();
}

}  // namespace amp
