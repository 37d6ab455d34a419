// amp_tile.hpp -- the fused trim + pileup kernel (variant 2), written for CDNA4 / gfx950.
//
// Work decomposition (nothing like the reference's per-read Python loop, A:896-915):
//   * a wave owns a TILE of 64 consecutive reads; a block of T_WAVES waves walks a contiguous
//     range of tiles of the coordinate-sorted batch, so it touches a bounded reference window
//   * per-position counters are PRIVATISED in LDS: win[6][W] uint32 for reference positions
//     [win_base, win_base+W); lanes add with LDS atomics (ds_add_u32) and the block flushes
//     the non-zero counters with global atomics when the window has to move / at the end.
//     Anything outside the window goes straight to the global table, so results never depend
//     on the input order -- only the speed does.
//   * the tile alternates between two lane mappings:
//       lane = read   P1  primer clips on the CIGAR held in LDS (one column per lane)
//                     P3  quality clip, outputs, classification, deletions / insertion events
//       lane = chunk  P2  sliding-window quality scan: 8 aligned bases per lane, window sums
//                         from a 16-byte neighbourhood, first failing window per read by
//                         LDS atomicMin/Max
//                     P4  base counting: 8 bases per lane (8 B of qual + 4 B of packed seq,
//                         coalesced), one LDS atomic per counted base.  The base order inside
//                         a chunk is rotated per lane so that the 32 lanes serviced together
//                         hit 32 different banks.
//   The match/mismatch bases of "regular" reads (clips only at the ends, body of M/=/X/I/D/N)
//   are counted by the chunk lanes; their deletions and insertion events are handled by the
//   read lane with a skip-ahead version of the exact pair walk.
//   * anything unusual is DEFERRED to the lane-per-read kernel (k_reads_deferred in
//     amplihip.hip), which runs the exact serial code of amp_read.hpp: reads with more CIGAR
//     ops than the LDS columns hold, reads of 64 k bases or more, reads whose trimmed CIGAR is
//     not regular, and (status only) regular reads on which a chunk lane met an error.
#pragma once

#include "amp_read.hpp"

namespace amp {

constexpr int TILE = 64;          // reads per wave tile
constexpr int T_WAVES = 8;        // waves per block
constexpr int T_W = 1024;         // reference positions covered by the LDS window
constexpr int T_MAXOPS = 8;       // CIGAR ops per read held in LDS (input ops <= T_MAXOPS-3)
constexpr int32_t NO_WINDOW = INT32_MIN;

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// entries of the deferred list: read index | kind
constexpr uint32_t DEFER_STATUS_ONLY = 0x80000000u;

// per-read state words kept in LDS for the chunk lanes
enum : int { S_POS, S_CB2, S_CB4, S_OFF8, S_LOHI, S_M, S_FF, S_INFO, S_WORDS };
// S_INFO bits
constexpr uint32_t I_REV = 1u, I_SIMPLE = 2u, I_NOCHUNK = 4u, I_ERRFLAG = 8u, I_INB = 16u;
constexpr int I_NCIG_SHIFT = 8;

struct WaveLds {
    uint32_t cigA[T_MAXOPS * TILE];
    uint32_t cigB[T_MAXOPS * TILE];
    uint32_t st[S_WORDS * TILE];
};
struct BlockLds {
    uint32_t win[AMP_NSYM * T_W];
    WaveLds wv[T_WAVES];
};

// one CIGAR column in LDS (stride TILE words)
struct LdsCig {
    lds_u32 *p;
    __device__ __forceinline__ uint32_t get(int i) const { return p[i * TILE]; }
    __device__ __forceinline__ void set(int i, uint32_t v) const { p[i * TILE] = v; }
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void lds_inc(lds_u32 *p) {
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    if (d < (uint32_t)T_W) lds_inc(t.win + col * T_W + d);
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
template <class CB, class Sink>
__device__ int count_regular_skip(const KParams &P, const CB &cig, int n, int32_t ref_start, int32_t lseq,
                                  int32_t qs, int32_t qe, const uint8_t *qual, Sink &sink) {
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
        if ((int32_t)qual[q] < mq) continue;
        if (q < qs) continue;
        if (q >= qe) break;
        const int32_t q0 = q;
        bool q_none = false;
        while (r < 0 && !q_none && q < qe) {
            if ((int32_t)qual[q] < mq) break;
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

// Classification of a final CIGAR.  regular: H* S* (M|=|X|I|D|N)* S* H* with query length ==
// lseq; simple: regular and the body holds match ops only, so query bases [m0, m1) map to
// reference positions ref_start + (q - m0).  For non-simple reads [m0, m1) spans from the
// first to the last match base.
template <class CB>
__device__ void classify(const CB &c, int n, int32_t lseq, bool &regular, bool &simple, int32_t &m0, int32_t &m1) {
    int phase = 0;  // 0 lead H, 1 lead S, 2 body, 3 trail S, 4 trail H
    int32_t q = 0, nm = 0, nother = 0;
    regular = true; m0 = m1 = 0;
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
            if (is_match_op(op)) {
                if (nm++ == 0) m0 = q;
                q += len;
                m1 = q;
            } else {
                ++nother;
                if (op == OP_I) q += len;
            }
        } else {
            regular = false; break;
        }
    }
    if (q != lseq) regular = false;
    simple = regular && nother == 0;
}

// A(1) C(2) G(4) T(8) -> 0..3, N(15) -> 4, anything else -> 15
__device__ __forceinline__ uint32_t col_of_code(uint32_t code) {
    const uint32_t lo = 0xFFF2F10Fu;  // codes 0..7
    const uint32_t hi = 0x4FFFFFF3u;  // codes 8..15
    uint32_t x = (code & 8u) ? hi : lo;
    return (x >> ((code & 7u) * 4u)) & 15u;
}

// bytes of a dword that lie in [kmin, kmax) when the dword holds byte indices [base, base+4)
__device__ __forceinline__ uint32_t byte_range_mask(int32_t kmin, int32_t kmax, int32_t base) {
    int32_t a = kmin - base, b = kmax - base;
    a = a < 0 ? 0 : (a > 4 ? 4 : a);
    b = b < 0 ? 0 : (b > 4 ? 4 : b);
    uint32_t ma = a >= 4 ? 0xFFFFFFFFu : ((1u << (a * 8)) - 1u);
    uint32_t mb = b >= 4 ? 0xFFFFFFFFu : ((1u << (b * 8)) - 1u);
    return mb & ~ma;
}

// sum of the W bytes starting at byte `b` (0..7) of the 16-byte group w[0..3]
template <int W>
__device__ __forceinline__ uint32_t window_sum_at(const uint32_t (&w)[4], int b) {
    // 8 bytes starting at byte b
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

template <int W>
__device__ __forceinline__ void window_sums8(const uint32_t (&w)[4], uint32_t (&s)[8]) {
#pragma unroll
    for (int b = 0; b < 8; ++b) s[b] = window_sum_at<W>(w, b);
}

__device__ __forceinline__ void window_sums8_dyn(int Wd, const uint32_t (&w)[4], uint32_t (&s)[8]) {
    switch (Wd) {
        case 1: window_sums8<1>(w, s); break; case 2: window_sums8<2>(w, s); break;
        case 3: window_sums8<3>(w, s); break; case 4: window_sums8<4>(w, s); break;
        case 5: window_sums8<5>(w, s); break; case 6: window_sums8<6>(w, s); break;
        case 7: window_sums8<7>(w, s); break; default: window_sums8<8>(w, s); break;
    }
}

__device__ __forceinline__ int find_read(const lds_u32 *cb, uint32_t c, float inv) {
    int r = (int)((float)c * inv);
    r = r > 63 ? 63 : r;
    while (c < cb[r]) --r;
    while (r < 63 && c >= cb[r + 1]) ++r;
    return r;
}

// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(T_WAVES * 64)
k_tile(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *counts, EventBuf eb, uint32_t *dlist,
       int tiles_per_block, uint32_t phases) {
    unsigned long long *const ctr = eb.ctr;
    __shared__ BlockLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t n = rd.n_reads;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_block;
    const int64_t tile_end = tile_begin + tiles_per_block < n_tiles ? tile_begin + tiles_per_block : n_tiles;
    if (tile_begin >= tile_end) return;

    lds_u32 *const win = (lds_u32 *)L.win;
    for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) win[i] = 0;
    int32_t win_base = NO_WINDOW;
    lds_u32 *const st = (lds_u32 *)L.wv[wave].st;
    lds_u32 *const cigA = (lds_u32 *)L.wv[wave].cigA;
    lds_u32 *const cigB = (lds_u32 *)L.wv[wave].cigB;
    TileCtx tc{win, 0, 0u, counts, eb, (uint32_t)P.ref_len};
    const int32_t mq = P.min_quality;
    unsigned long long n_err = 0;

    for (int64_t t0 = tile_begin; t0 < tile_end; t0 += T_WAVES) {
        // ---- window management (uniform over the block) ------------------------------------
        const int32_t first_pos = rd.pos[t0 * TILE];
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
            __syncthreads();
        }
        tc.win_base = win_base;
        {
            int64_t lim = (int64_t)P.ref_len - win_base;
            tc.wlim = lim <= 0 ? 0u : (lim > T_W ? (uint32_t)T_W : (uint32_t)lim);
        }
        const int64_t tile = t0 + wave;
        if (tile >= tile_end) continue;

        // =================================== P1: lane = read ===================================
        const int64_t i = tile * TILE + lane;
        const bool valid = i < n;
        int32_t lseq = 0, pos = 0, tlen = 0;
        uint32_t flag = 0, c0 = 0, off8 = 0;
        int ncig = 0;
        if (valid) {
            pos = rd.pos[i]; flag = rd.flag[i]; tlen = rd.tlen[i]; lseq = (int32_t)rd.lseq[i];
            c0 = rd.cig_off32[i]; ncig = (int)(rd.cig_off32[i + 1] - c0); off8 = rd.seq_off8[i];
        }
        const size_t slot = (size_t)c0 + 3 * (size_t)(valid ? i : 0);
        const int64_t boff = (int64_t)off8 * 8;
        const uint8_t *qual = rd.qual + boff;
        bool defer_full = valid && (ncig + 3 > T_MAXOPS || (uint32_t)lseq >= 65536u);
        const bool mine = valid && !defer_full;
        const bool have_qual = mine && lseq > 0 && qual[0] != 0xFF;
        TrimState ts{pos, ncig, 0u, 0};
        LdsCig cur{cigA + lane}, tmp{cigB + lane};
        int32_t qs = 0, lo = 0, qlen = 0;
        bool can_q = false;
        if (mine) {
            for (int k = 0; k < ncig; ++k) cur.set(k, rd.cig[c0 + k]);
            if (P.do_trim) {
                trim_primers(P, ts, flag, tlen, lseq, cur, tmp);
                if (!ts.err) can_q = quality_window(ts, lseq, have_qual, cur, qs, lo, qlen);
            }
        }
        const bool rev = (flag & 0x10u) != 0;
        uint32_t nch2 = (mine && can_q) ? (uint32_t)(((lo + qlen + 7) >> 3) - (lo >> 3)) : 0u;
        uint32_t incl = nch2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const uint32_t total2 = __shfl(incl, 63);
        st[S_CB2 * TILE + lane] = incl - nch2;
        st[S_LOHI * TILE + lane] = (uint32_t)lo | ((uint32_t)(lo + qlen) << 16);
        st[S_OFF8 * TILE + lane] = off8;
        st[S_FF * TILE + lane] = rev ? 0u : (uint32_t)qlen;
        st[S_INFO * TILE + lane] = (rev ? I_REV : 0u);
        wave_sync();

        // =================================== P2: lane = chunk ===================================
        if (P.do_trim && (phases & 2u)) {
            const float inv2 = total2 ? 64.0f / (float)total2 : 0.0f;
            const int32_t Wd = P.window;
            for (uint32_t c = lane; c < total2; c += 64) {
                const int r = find_read(st + S_CB2 * TILE, c, inv2);
                const uint32_t lohi = st[S_LOHI * TILE + r];
                const int32_t rlo = (int32_t)(lohi & 0xFFFFu), rhi = (int32_t)(lohi >> 16);
                const int32_t j0 = ((int32_t)(c - st[S_CB2 * TILE + r]) + (rlo >> 3)) * 8;
                const uint8_t *qp = rd.qual + (int64_t)st[S_OFF8 * TILE + r] * 8;
                const bool rrev = st[S_INFO * TILE + r] & I_REV;
                if (Wd <= 8) {
                    // 16-byte neighbourhood: [j0-8, j0+8) for reverse reads, [j0, j0+16) for forward
                    const int32_t a0 = rrev ? j0 - 8 : j0;
                    uint2 w0 = make_uint2(0, 0), w1 = make_uint2(0, 0);
                    if (a0 >= 0) w0 = *(const uint2 *)(qp + a0);
                    if (a0 + 8 < rhi) w1 = *(const uint2 *)(qp + a0 + 8);
                    // zero the bytes outside [rlo, rhi)
                    const int32_t kmin = rlo - a0, kmax = rhi - a0;
                    uint32_t w[4] = {w0.x, w0.y, w1.x, w1.y};
                    if (kmin > 0 || kmax < 16) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) w[d] &= byte_range_mask(kmin, kmax, d * 4);
                    }
                    uint32_t s[8];
                    if (rrev) {
                        // mirror the 16 bytes so that the window of a base runs towards higher indices:
                        // mirrored byte k = base a0+15-k; base j0+7-bb is mirrored byte bb
                        uint32_t m[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) m[d] = __builtin_bswap32(w[3 - d]);
                        window_sums8_dyn(Wd, m, s);
                        int32_t best = 0;
#pragma unroll
                        for (int bb = 7; bb >= 0; --bb) {
                            const int32_t a = j0 + 7 - bb;          // last base of the window
                            const int32_t iend = a + 1 - rlo;       // the reference's loop variable i
                            const int32_t wl_ = iend < Wd ? iend : Wd;
                            if (a >= rlo && a < rhi && (int32_t)s[bb] < mq * wl_) best = iend;
                        }
                        if (best) __hip_atomic_fetch_max(st + S_FF * TILE + r, (uint32_t)best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        window_sums8_dyn(Wd, w, s);
                        int32_t best = -1;
#pragma unroll
                        for (int bb = 7; bb >= 0; --bb) {
                            const int32_t a = j0 + bb;
                            const int32_t left = rhi - a;
                            const int32_t wl_ = left < Wd ? left : Wd;
                            if (a >= rlo && a < rhi && (int32_t)s[bb] < mq * wl_) best = a - rlo;
                        }
                        if (best >= 0) __hip_atomic_fetch_min(st + S_FF * TILE + r, (uint32_t)best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    // wide windows: direct sums from global memory (rare parameter choice)
                    for (int bb = 0; bb < 8; ++bb) {
                        const int32_t a = j0 + bb;
                        if (a < rlo || a >= rhi) continue;
                        if (rrev) {
                            const int32_t iend = a + 1 - rlo;
                            const int32_t wl_ = iend < Wd ? iend : Wd;
                            int64_t sum = 0;
                            for (int32_t k = 0; k < wl_; ++k) sum += qp[a - k];
                            if (sum < (int64_t)mq * wl_) __hip_atomic_fetch_max(st + S_FF * TILE + r, (uint32_t)iend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
                            const int32_t left = rhi - a;
                            const int32_t wl_ = left < Wd ? left : Wd;
                            int64_t sum = 0;
                            for (int32_t k = 0; k < wl_; ++k) sum += qp[a + k];
                            if (sum < (int64_t)mq * wl_) __hip_atomic_fetch_min(st + S_FF * TILE + r, (uint32_t)(a - rlo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            wave_sync();
        }

        // =================================== P3: lane = read ===================================
        int cerr = 0;
        bool chunks = false, simple = false;
        int32_t m0 = 0, m1 = 0;
        if (mine && !ts.err && P.do_trim && can_q) {
            const int32_t ff = (int32_t)st[S_FF * TILE + lane];
            trim_quality_apply(ts, rev, ff, qlen, qs, cur, tmp);
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
        if (mine && !ts.err && P.do_count) {
            bool regular;
            classify(cur, ts.n, lseq, regular, simple, m0, m1);
            if (!have_qual) regular = false;
            int e1 = 0, e2 = 0;
            int32_t fqs = 0, fqe = 0;
            if (regular) { fqs = query_alignment_start(cur, ts.n, lseq, e1); fqe = query_alignment_end(cur, ts.n, lseq, e2); }
            if (!regular || e1 || e2) {
                defer_full = true;
            } else {
                chunks = m1 > m0;
                if (!simple && (phases & 8u)) {
                    TileSink sink{tc, (uint32_t)(read_base + (uint64_t)i)};
                    cerr = count_regular_skip(P, cur, ts.n, ts.pos, lseq, fqs, fqe, qual, sink);
                }
            }
        }
        uint32_t nch4 = chunks ? (uint32_t)(((m1 + 7) >> 3) - (m0 >> 3)) : 0u;
        incl = nch4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const uint32_t total4 = __shfl(incl, 63);
        st[S_CB4 * TILE + lane] = incl - nch4;
        st[S_M * TILE + lane] = (uint32_t)m0 | ((uint32_t)m1 << 16);
        st[S_POS * TILE + lane] = (uint32_t)ts.pos;
        st[S_INFO * TILE + lane] = (simple ? I_SIMPLE : 0u) | ((cur.p != cigA + lane) ? I_INB : 0u) | ((uint32_t)ts.n << I_NCIG_SHIFT);
        wave_sync();

        // =================================== P4: lane = chunk ===================================
        if (P.do_count && (phases & 4u)) {
            const float inv4 = total4 ? 64.0f / (float)total4 : 0.0f;
            for (uint32_t c = lane; c < total4; c += 64) {
                const int r = find_read(st + S_CB4 * TILE, c, inv4);
                const uint32_t rinfo = st[S_INFO * TILE + r];
                const uint32_t mm = st[S_M * TILE + r];
                const int32_t rm0 = (int32_t)(mm & 0xFFFFu), rm1 = (int32_t)(mm >> 16);
                const int32_t j0 = ((int32_t)(c - st[S_CB4 * TILE + r]) + (rm0 >> 3)) * 8;
                const int64_t rb = (int64_t)st[S_OFF8 * TILE + r] * 8;
                const uint2 qw = *(const uint2 *)(rd.qual + rb + j0);
                uint32_t sw = *(const uint32_t *)(rd.seq + ((rb + j0) >> 1));
                const int32_t rpos = (int32_t)st[S_POS * TILE + r];
                bool bad = false;
                if (rinfo & I_SIMPLE) {
                    // rotate the 8 bases by `rot` so that lanes serviced together spread over banks
                    const uint32_t rot = ((uint32_t)lane >> 2) & 7u;
                    uint32_t qa = (rot & 4u) ? qw.y : qw.x, qb = (rot & 4u) ? qw.x : qw.y;
                    const uint32_t sh = (rot & 3u) * 8u;
                    const uint32_t q0w = __builtin_amdgcn_alignbit(qb, qa, sh);   // bytes rot..rot+3
                    const uint32_t q1w = __builtin_amdgcn_alignbit(qa, qb, sh);   // bytes rot+4..rot+7 (mod 8)
                    sw = ((sw & 0x0F0F0F0Fu) << 4) | ((sw >> 4) & 0x0F0F0F0Fu);   // base k at bits [4k, 4k+4)
                    sw = __builtin_amdgcn_alignbit(sw, sw, rot * 4u);
                    const int32_t d0 = rpos + (j0 - rm0) - win_base;               // window offset of base 0
                    const uint32_t span = (uint32_t)(rm1 - rm0);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const uint32_t bb = ((uint32_t)b + rot) & 7u;
                        const uint32_t qv = ((b < 4 ? q0w : q1w) >> ((b & 3) * 8)) & 0xFFu;
                        const uint32_t code = (sw >> (b * 4)) & 15u;
                        const uint32_t qrel = (uint32_t)(j0 - rm0) + bb;
                        if (qrel < span && (int32_t)qv >= mq) {
                            const uint32_t col = col_of_code(code);
                            const uint32_t d = (uint32_t)d0 + bb;
                            if (col > 4u) bad = true;
                            else if (d < tc.wlim) lds_inc(win + col * T_W + d);
                            else {
                                const int32_t rp = win_base + (int32_t)d;
                                if ((uint32_t)rp >= tc.G) bad = true;
                                else atomicAdd(&counts[(size_t)rp * AMP_NSYM + col], 1u);
                            }
                        }
                    }
                } else {
                    // walk the read's ops to the chunk, then base by base
                    const int ncg = (int)(rinfo >> I_NCIG_SHIFT);
                    LdsCig cg{((rinfo & I_INB) ? cigB : cigA) + r};
                    int k = 0;
                    int32_t qq = 0, rr = rpos, oplen = 0, opq = 0;
                    uint32_t op = OP_H;
                    for (; k < ncg; ++k) {
                        uint32_t v = cg.get(k); op = v & 15u; oplen = (int32_t)(v >> 4);
                        if (op == OP_H) { oplen = 0; continue; }
                        if (consumes_query(op)) {
                            if (j0 < qq + oplen) { opq = qq; break; }
                            qq += oplen;
                        }
                        if (consumes_ref(op)) rr += oplen;
                    }
                    if (k < ncg) {
                        int32_t q = j0;
                        if (consumes_ref(op)) rr += q - opq;
                        int32_t left = opq + oplen - q;
                        for (int b = 0; b < 8; ++b, ++q) {
                            while (left == 0) {
                                ++k;
                                if (k >= ncg) break;
                                uint32_t v = cg.get(k); op = v & 15u; oplen = (int32_t)(v >> 4);
                                if (op == OP_H) continue;
                                if (consumes_query(op)) left = oplen; else if (consumes_ref(op)) rr += oplen;
                            }
                            if (k >= ncg) break;
                            --left;
                            if (is_match_op(op)) {
                                const uint32_t qv = ((b & 4 ? qw.y : qw.x) >> ((b & 3) * 8)) & 0xFFu;
                                if ((int32_t)qv >= mq) {
                                    const uint32_t byte = (sw >> ((b >> 1) * 8)) & 0xFFu;
                                    const uint32_t code = (b & 1) ? (byte & 15u) : (byte >> 4);
                                    const uint32_t col = col_of_code(code);
                                    if (col > 4u || (uint32_t)rr >= tc.G) bad = true;
                                    else tile_add(tc, rr, col);
                                }
                                ++rr;
                            }
                        }
                    }
                }
                if (bad) __hip_atomic_fetch_or(st + S_INFO * TILE + r, I_ERRFLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_sync();
        }

        // ---- status / deferral (lane = read) -----------------------------------------------------
        if (valid) {
            uint32_t status = (uint32_t)ts.err;
            if (defer_full) {
                dlist[atomicAdd(&ctr[3], 1ull)] = (uint32_t)i;
                status = 0;   // overwritten by the deferred kernel
            } else if (!status && P.do_count && (cerr || (st[S_INFO * TILE + lane] & I_ERRFLAG))) {
                dlist[atomicAdd(&ctr[3], 1ull)] = (uint32_t)i | DEFER_STATUS_ONLY;
            } else if (status) {
                ++n_err;
            }
            if (out.status) out.status[i] = (uint8_t)status;
        }
        wave_sync();
    }

    // ---- final flush ------------------------------------------------------------------------
    __syncthreads();
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
}

static inline int tile_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out,
                              uint32_t *counts, const EventBuf &eb, uint32_t *dlist, int n_cu, uint32_t phases,
                              hipStream_t stream) {
    const int64_t n_tiles = (rd.n_reads + TILE - 1) / TILE;
    if (n_tiles == 0) return 0;
    int64_t max_blocks = (int64_t)n_cu * 2;
    int64_t tpb = (n_tiles + max_blocks - 1) / max_blocks;
    tpb = ((tpb + T_WAVES - 1) / T_WAVES) * T_WAVES;   // whole super-tiles per block
    int64_t grid = (n_tiles + tpb - 1) / tpb;
    k_tile<<<(unsigned)grid, T_WAVES * 64, 0, stream>>>(P, rd, read_base, out, counts, eb, dlist, (int)tpb, phases);
    return (int)hipGetLastError();
}

}  // namespace amp
