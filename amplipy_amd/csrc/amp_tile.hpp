// amp_tile.hpp -- the fused trim + pileup kernel (variant 2), written for CDNA4 / gfx950.
//
// Work decomposition (nothing like the reference's per-read Python loop, A:896-915):
//   * a wave owns a TILE of 64 consecutive reads; a block of WAVES waves walks a contiguous
//     range of tiles of the coordinate-sorted batch, so it touches a bounded reference window
//   * per-position counters are PRIVATISED in LDS: win[6][W] uint32 for reference positions
//     [win_base, win_base+W); lanes add with LDS atomics (ds_add_u32) and the block flushes
//     the non-zero counters with global atomics when the window has to move / at the end.
//     Anything outside the window goes straight to the global table, so results never depend
//     on the input order -- only the speed does.
//   * the tile alternates between two lane mappings:
//       lane = read   P1  primer clips on the CIGAR held in LDS (one column per lane)
//                     P3  quality clip, outputs, classification, deletions / insertion events
//                     P5  status (exact first error for the rare failing read)
//       lane = chunk  P2  sliding-window quality scan: 8 aligned bases per lane, window sums
//                         from a 16-byte neighbourhood, first failing window per read by
//                         LDS atomicMin/Max
//                     P4  base counting: 8 bases per lane (8 B of qual + 4 B of packed seq,
//                         coalesced), one LDS atomic per counted base.  The base order inside
//                         a chunk is rotated per lane so that the 32 lanes serviced together
//                         hit 32 different banks.
//   The match/mismatch bases of "regular" reads (clips only at the ends, body of M/=/X/I/D/N)
//   are counted by the chunk lanes; their deletions and insertion events, and every read
//   with an unusual CIGAR, go through the exact pair walk of amp_read.hpp.
#pragma once

#include "amp_read.hpp"

namespace amp {

constexpr int TILE = 64;          // reads per wave tile
constexpr int T_WAVES = 8;        // waves per block
constexpr int T_W = 1024;         // reference positions covered by the LDS window
constexpr int T_MAXOPS = 6;       // CIGAR ops per read held in LDS (input ops <= T_MAXOPS-3)
constexpr int32_t NO_WINDOW = INT32_MIN;

// per-read state words kept in LDS for the chunk lanes
enum : int { S_POS, S_LO, S_HI, S_QS, S_CB, S_OFF8, S_FF, S_INFO, S_M0, S_M1, S_R0, S_NCIG, S_SLOT, S_WORDS };
// S_INFO bits
constexpr uint32_t I_REV = 1u, I_REGULAR = 2u, I_SIMPLE = 4u, I_NOCHUNK = 8u, I_GLOBAL = 16u, I_ERRFLAG = 32u,
                   I_INB = 64u /* final CIGAR lives in LDS buffer B */;

struct WaveLds {
    uint32_t cigA[T_MAXOPS * TILE];
    uint32_t cigB[T_MAXOPS * TILE];
    uint32_t st[S_WORDS * TILE];
};
struct BlockLds {
    uint32_t win[AMP_NSYM * T_W];
    WaveLds wv[T_WAVES];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct TileCtx {
    uint32_t *win;
    int32_t win_base;
    uint32_t *counts;
    amp_ins_event *ev;
    unsigned long long *ctr;
    long long ev_cap;
    uint32_t G;
};

__device__ __forceinline__ void tile_add(const TileCtx &t, int32_t r, uint32_t col) {
    uint32_t d = (uint32_t)(r - t.win_base);
    if (d < (uint32_t)T_W) atomicAdd(&t.win[col * T_W + d], 1u);
    else atomicAdd(&t.counts[(size_t)r * AMP_NSYM + col], 1u);
}

struct TileSink {   // exact walk, real effects
    const TileCtx &t;
    uint32_t read;
    __device__ void add(int32_t r, uint32_t col) { tile_add(t, r, col); }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
        unsigned long long idx = atomicAdd(&t.ctr[0], 1ull);
        if ((long long)idx < t.ev_cap) t.ev[idx] = amp_ins_event{pos, read, lo, hi};
    }
};
struct NullSink {   // dry run: only the status matters
    __device__ void add(int32_t, uint32_t) {}
    __device__ void event(int32_t, int32_t, int32_t) {}
};

// CIGAR access for chunk lanes: either LDS column `r` or the read's global output slot.
struct CigRef {
    const uint32_t *lds;   // &buf[r], stride TILE
    const uint32_t *glb;   // slot base, stride 1 (null when in LDS)
    __device__ __forceinline__ uint32_t get(int k) const { return glb ? glb[k] : lds[k * TILE]; }
};

// The skip-ahead variant of the exact walk for REGULAR reads: M/=/X runs and the end clips
// are stepped over in O(1) (their bases are counted by the chunk lanes / have no effect);
// deletions, reference skips and insertion runs take exactly the path of count_read_walk.
template <int S, class Sink>
__device__ int count_regular_skip(const KParams &P, const CigBuf<S> &cig, int n, int32_t ref_start, int32_t lseq,
                                  int32_t qs, int32_t qe, const uint8_t *qual, Sink &sink) {
    const int32_t ref_end = ref_start + reference_length(cig, n);
    const uint32_t G = (uint32_t)P.ref_len;
    const int32_t mq = P.min_quality;
    PairIter<S> it;
    it.init(cig, n, ref_start);
    int32_t q, r;
    bool pending = false;
    int32_t pend_q = 0, pend_r = 0;
    for (;;) {
        if (pending) {
            q = pend_q; r = pend_r; pending = false;
        } else {
            // step over whole ops that cannot produce effects here
            while (it.j >= it.len) {
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
            // a match base handed back by an insertion scan: the chunk lanes count it; skip the
            // rest of its op
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
// reference positions ref_start + (q - m0).
template <int S>
__device__ void classify(const CigBuf<S> &c, int n, int32_t lseq, bool &regular, bool &simple, int32_t &m0, int32_t &m1) {
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

__device__ __forceinline__ uint32_t col_of_code(uint32_t code) {
    // A(1) C(2) G(4) T(8) -> 0..3, N(15) -> 4, anything else -> 15
    const uint32_t lo = 0xFFF2F10Fu;  // codes 0..7
    const uint32_t hi = 0x4FFFFFF3u;  // codes 8..15
    uint32_t x = (code & 8u) ? hi : lo;
    return (x >> ((code & 7u) * 4u)) & 15u;
}

template <int W>
__device__ __forceinline__ void window_sums(const uint32_t (&t)[16], uint32_t (&s)[8]) {
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) acc += t[k];
    s[0] = acc;
#pragma unroll
    for (int b = 1; b < 8; ++b) { acc = acc - t[b - 1] + t[b - 1 + W]; s[b] = acc; }
}

// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(T_WAVES * 64)
k_tile(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *gscratch, uint32_t *counts,
       amp_ins_event *ev, unsigned long long *ctr, long long ev_cap, int tiles_per_block) {
    __shared__ BlockLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t n = rd.n_reads;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_block;
    const int64_t tile_end = tile_begin + tiles_per_block < n_tiles ? tile_begin + tiles_per_block : n_tiles;
    if (tile_begin >= tile_end) return;

    for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) L.win[i] = 0;
    int32_t win_base = NO_WINDOW;
    WaveLds &wl = L.wv[wave];
    uint32_t *const st = wl.st;
    TileCtx tc{L.win, 0, counts, ev, ctr, ev_cap, (uint32_t)P.ref_len};
    const int32_t mq = P.min_quality;
    unsigned long long n_err = 0;

    for (int64_t t0 = tile_begin; t0 < tile_end; t0 += T_WAVES) {
        // ---- window management (uniform over the block) ------------------------------------
        const int32_t first_pos = rd.pos[t0 * TILE];
        if (win_base == NO_WINDOW || first_pos < win_base || first_pos - win_base >= T_W / 2) {
            __syncthreads();
            if (win_base != NO_WINDOW) {
                for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) {
                    uint32_t v = L.win[i];
                    if (v) {
                        int sym = i / T_W, d = i - sym * T_W;
                        atomicAdd(&counts[(size_t)(win_base + d) * AMP_NSYM + sym], v);
                        L.win[i] = 0;
                    }
                }
            }
            win_base = first_pos & ~31;
            __syncthreads();
        }
        tc.win_base = win_base;
        const int64_t tile = t0 + wave;
        if (tile >= tile_end) continue;

        // =================================== P1: lane = read ===================================
        const int64_t i = tile * TILE + lane;
        const bool valid = i < n;
        int32_t lseq = 0, pos = 0;
        uint32_t flag = 0, c0 = 0, off8 = 0;
        int ncig = 0;
        int32_t tlen = 0;
        if (valid) {
            pos = rd.pos[i]; flag = rd.flag[i]; tlen = rd.tlen[i]; lseq = (int32_t)rd.lseq[i];
            c0 = rd.cig_off32[i]; ncig = (int)(rd.cig_off32[i + 1] - c0); off8 = rd.seq_off8[i];
        }
        const size_t slot = (size_t)c0 + 3 * (size_t)(valid ? i : 0);
        const bool use_glb = valid && (ncig + 3 > T_MAXOPS);
        const int64_t boff = (int64_t)off8 * 8;
        const uint8_t *qual = rd.qual + boff;
        const bool have_qual = valid && lseq > 0 && qual[0] != 0xFF;
        TrimState ts{pos, ncig, 0u, 0};
        CigBuf<TILE> lcur{wl.cigA + lane}, ltmp{wl.cigB + lane};
        CigBuf<1> gcur{out.new_cig + slot}, gtmp{gscratch + slot};
        uint32_t *const ghome = gcur.p;
        int32_t qs = 0, lo = 0, qlen = 0;
        bool can_q = false;
        if (valid) {
            if (use_glb) {
                for (int k = 0; k < ncig; ++k) gcur.set(k, rd.cig[c0 + k]);
                if (P.do_trim) { trim_primers(P, ts, flag, tlen, lseq, gcur, gtmp); if (!ts.err) can_q = quality_window(ts, lseq, have_qual, gcur, qs, lo, qlen); }
            } else {
                for (int k = 0; k < ncig; ++k) lcur.set(k, rd.cig[c0 + k]);
                if (P.do_trim) { trim_primers(P, ts, flag, tlen, lseq, lcur, ltmp); if (!ts.err) can_q = quality_window(ts, lseq, have_qual, lcur, qs, lo, qlen); }
            }
        }
        const bool rev = (flag & 0x10u) != 0;
        // chunk range of this read: stored bases [lo, lo+qlen) when trimming, everything otherwise
        int32_t ch_lo = 0, ch_hi = 0;
        if (valid && !ts.err) {
            if (P.do_trim) { if (can_q) { ch_lo = lo >> 3; ch_hi = (lo + qlen + 7) >> 3; } }
            else { ch_lo = 0; ch_hi = (lseq + 7) >> 3; }
        }
        uint32_t nch = (uint32_t)(ch_hi - ch_lo);
        uint32_t incl = nch;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const uint32_t total_ch = __shfl(incl, 63);
        st[S_CB * TILE + lane] = incl - nch;
        st[S_LO * TILE + lane] = (uint32_t)lo;
        st[S_HI * TILE + lane] = (uint32_t)(lo + qlen);
        st[S_OFF8 * TILE + lane] = off8;
        st[S_FF * TILE + lane] = rev ? 0u : (uint32_t)qlen;
        st[S_INFO * TILE + lane] = (rev ? I_REV : 0u) | ((valid && !ts.err && can_q) ? 0u : I_NOCHUNK);
        st[S_QS * TILE + lane] = (uint32_t)ch_lo;
        wave_sync();

        // =================================== P2: lane = chunk ===================================
        const float inv = total_ch ? 64.0f / (float)total_ch : 0.0f;
        if (P.do_trim) {
            const int32_t Wd = P.window;
            for (uint32_t c = lane; c < total_ch; c += 64) {
                int r = (int)((float)c * inv);
                r = r > 63 ? 63 : r;
                while (c < st[S_CB * TILE + r]) --r;
                while (r < 63 && c >= st[S_CB * TILE + r + 1]) ++r;
                const uint32_t info = st[S_INFO * TILE + r];
                if (info & I_NOCHUNK) continue;
                const int32_t rlo = (int32_t)st[S_LO * TILE + r], rhi = (int32_t)st[S_HI * TILE + r];
                const int32_t j0 = ((int32_t)(c - st[S_CB * TILE + r]) + (int32_t)st[S_QS * TILE + r]) * 8;
                const uint8_t *qp = rd.qual + (int64_t)st[S_OFF8 * TILE + r] * 8;
                const bool rrev = info & I_REV;
                if (Wd <= 8) {
                    // 16-byte neighbourhood: [j0-8, j0+8) for reverse reads, [j0, j0+16) for forward
                    const int32_t a0 = rrev ? j0 - 8 : j0;
                    uint2 w0 = make_uint2(0, 0), w1 = make_uint2(0, 0);
                    if (a0 >= 0) w0 = *(const uint2 *)(qp + a0);
                    if (a0 + 8 < rhi && a0 + 8 >= 0) w1 = *(const uint2 *)(qp + a0 + 8);
                    uint32_t t[16];
                    const uint32_t ww[4] = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        int32_t a = a0 + k;
                        uint32_t v = (ww[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                        t[k] = (a >= rlo && a < rhi) ? v : 0u;
                    }
                    uint32_t s[8];
                    if (rrev) {
                        // window ending at base b = a0+8+bb covers t[bb+9-W .. bb+8]
                        uint32_t tt[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) tt[k] = t[15 - k];   // mirror: base a0+15-k
                        // after mirroring, base j0+7-bb is tt[bb]; its window is tt[bb .. bb+W-1]
                        switch (Wd) {
                            case 1: window_sums<1>(tt, s); break; case 2: window_sums<2>(tt, s); break;
                            case 3: window_sums<3>(tt, s); break; case 4: window_sums<4>(tt, s); break;
                            case 5: window_sums<5>(tt, s); break; case 6: window_sums<6>(tt, s); break;
                            case 7: window_sums<7>(tt, s); break; default: window_sums<8>(tt, s); break;
                        }
                        int32_t best = 0;
#pragma unroll
                        for (int bb = 0; bb < 8; ++bb) {
                            const int32_t a = j0 + 7 - bb;              // absolute query index of the window's last base
                            const int32_t iend = a + 1 - rlo;           // the reference's loop variable i
                            const int32_t wl_ = iend < Wd ? iend : Wd;
                            if (a >= rlo && a < rhi && (int64_t)s[bb] < (int64_t)mq * wl_) best = best > iend ? best : iend;
                        }
                        if (best) atomicMax(&st[S_FF * TILE + r], (uint32_t)best);
                    } else {
                        switch (Wd) {
                            case 1: window_sums<1>(t, s); break; case 2: window_sums<2>(t, s); break;
                            case 3: window_sums<3>(t, s); break; case 4: window_sums<4>(t, s); break;
                            case 5: window_sums<5>(t, s); break; case 6: window_sums<6>(t, s); break;
                            case 7: window_sums<7>(t, s); break; default: window_sums<8>(t, s); break;
                        }
                        int32_t best = INT32_MAX;
#pragma unroll
                        for (int bb = 7; bb >= 0; --bb) {
                            const int32_t a = j0 + bb;
                            const int32_t left = rhi - a;
                            const int32_t wl_ = left < Wd ? left : Wd;
                            if (a >= rlo && a < rhi && (int64_t)s[bb] < (int64_t)mq * wl_) best = a - rlo;
                        }
                        if (best != INT32_MAX) atomicMin(&st[S_FF * TILE + r], (uint32_t)best);
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
                            if (sum < (int64_t)mq * wl_) atomicMax(&st[S_FF * TILE + r], (uint32_t)iend);
                        } else {
                            const int32_t left = rhi - a;
                            const int32_t wl_ = left < Wd ? left : Wd;
                            int64_t sum = 0;
                            for (int32_t k = 0; k < wl_; ++k) sum += qp[a + k];
                            if (sum < (int64_t)mq * wl_) atomicMin(&st[S_FF * TILE + r], (uint32_t)(a - rlo));
                        }
                    }
                }
            }
            wave_sync();
        }

        // =================================== P3: lane = read ===================================
        int cerr = 0;          // counting status of this lane's read
        bool nochunk = true;
        if (valid && !ts.err) {
            if (P.do_trim && can_q) {
                const int32_t ff = (int32_t)st[S_FF * TILE + lane];
                if (use_glb) trim_quality_apply(ts, rev, ff, qlen, qs, gcur, gtmp);
                else trim_quality_apply(ts, rev, ff, qlen, qs, lcur, ltmp);
            }
        }
        int32_t reflen = 0;
        if (valid && !ts.err) {
            if (use_glb) {
                if (gcur.p != ghome) { for (int k = 0; k < ts.n; ++k) ghome[k] = gcur.get(k); gcur.p = ghome; }
                reflen = reference_length(gcur, ts.n);
            } else {
                for (int k = 0; k < ts.n; ++k) ghome[k] = lcur.get(k);
                reflen = reference_length(lcur, ts.n);
            }
        }
        if (valid) {
            if (out.new_pos) out.new_pos[i] = ts.pos;
            if (out.new_ncig) out.new_ncig[i] = ts.err ? 0u : (uint32_t)ts.n;
            if (out.ref_len) out.ref_len[i] = ts.err ? 0 : reflen;
            if (out.trim_flags) out.trim_flags[i] = ts.err ? (uint8_t)0 : (uint8_t)ts.flags;
        }
        uint32_t info = rev ? I_REV : 0u;
        if (valid && !ts.err && P.do_count) {
            bool regular, simple;
            int32_t m0, m1;
            if (use_glb) classify(gcur, ts.n, lseq, regular, simple, m0, m1);
            else classify(lcur, ts.n, lseq, regular, simple, m0, m1);
            if (!have_qual || lseq == 0) regular = false;
            if (regular) {
                int e1 = 0, e2 = 0;
                int32_t fqs, fqe;
                if (use_glb) { fqs = query_alignment_start(gcur, ts.n, lseq, e1); fqe = query_alignment_end(gcur, ts.n, lseq, e2); }
                else { fqs = query_alignment_start(lcur, ts.n, lseq, e1); fqe = query_alignment_end(lcur, ts.n, lseq, e2); }
                if (e1 || e2) regular = false;
                else {
                    nochunk = false;
                    info |= I_REGULAR | (simple ? I_SIMPLE : 0u) | (use_glb ? I_GLOBAL : 0u);
                    st[S_M0 * TILE + lane] = (uint32_t)m0;
                    st[S_M1 * TILE + lane] = (uint32_t)m1;
                    st[S_R0 * TILE + lane] = (uint32_t)ts.pos;
                    if (!simple) {
                        TileSink sink{tc, (uint32_t)(read_base + (uint64_t)i)};
                        if (use_glb) cerr = count_regular_skip(P, gcur, ts.n, ts.pos, lseq, fqs, fqe, qual, sink);
                        else cerr = count_regular_skip(P, lcur, ts.n, ts.pos, lseq, fqs, fqe, qual, sink);
                    }
                }
            }
            if (!regular) {
                TileSink sink{tc, (uint32_t)(read_base + (uint64_t)i)};
                if (use_glb) cerr = count_read_walk(P, gcur, ts.n, ts.pos, lseq, rd.seq, boff, qual, have_qual, sink);
                else cerr = count_read_walk(P, lcur, ts.n, ts.pos, lseq, rd.seq, boff, qual, have_qual, sink);
            }
        }
        if (nochunk) info |= I_NOCHUNK;
        if (!use_glb && lcur.p != wl.cigA + lane) info |= I_INB;
        st[S_INFO * TILE + lane] = info;
        st[S_POS * TILE + lane] = (uint32_t)ts.pos;
        st[S_NCIG * TILE + lane] = (uint32_t)ts.n;
        st[S_SLOT * TILE + lane] = (uint32_t)0;
        wave_sync();

        // =================================== P4: lane = chunk ===================================
        if (P.do_count) {
            for (uint32_t c = lane; c < total_ch; c += 64) {
                int r = (int)((float)c * inv);
                r = r > 63 ? 63 : r;
                while (c < st[S_CB * TILE + r]) --r;
                while (r < 63 && c >= st[S_CB * TILE + r + 1]) ++r;
                const uint32_t rinfo = st[S_INFO * TILE + r];
                if (rinfo & I_NOCHUNK) continue;
                const int32_t cidx = (int32_t)(c - st[S_CB * TILE + r]) + (int32_t)st[S_QS * TILE + r];
                const int32_t j0 = cidx * 8;
                const int64_t rb = (int64_t)st[S_OFF8 * TILE + r] * 8;
                const uint2 qw = *(const uint2 *)(rd.qual + rb + j0);
                const uint32_t sw = *(const uint32_t *)(rd.seq + ((rb + j0) >> 1));
                const uint32_t rot = ((uint32_t)lane >> 2) & 7u;
                bool bad = false;
                if (rinfo & I_SIMPLE) {
                    const int32_t m0 = (int32_t)st[S_M0 * TILE + r], m1 = (int32_t)st[S_M1 * TILE + r];
                    const int32_t r0 = (int32_t)st[S_R0 * TILE + r];
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const uint32_t bb = ((uint32_t)b + rot) & 7u;
                        const int32_t q = j0 + (int32_t)bb;
                        const uint32_t qv = ((bb & 4u ? qw.y : qw.x) >> ((bb & 3u) * 8u)) & 0xFFu;
                        // packed seq: byte bb>>1 of sw, high nibble first
                        const uint32_t byte = (sw >> ((bb >> 1) * 8u)) & 0xFFu;
                        const uint32_t code = (bb & 1u) ? (byte & 15u) : (byte >> 4);
                        if (q >= m0 && q < m1 && (int32_t)qv >= mq) {
                            const int32_t rp = r0 + (q - m0);
                            const uint32_t col = col_of_code(code);
                            if (col > 4u || (uint32_t)rp >= tc.G) bad = true;
                            else tile_add(tc, rp, col);
                        }
                    }
                } else {
                    // walk the read's ops to the chunk, then base by base
                    const int ncg = (int)st[S_NCIG * TILE + r];
                    CigRef cg;
                    if (rinfo & I_GLOBAL) {
                        const int64_t ri = tile * TILE + r;
                        cg.glb = out.new_cig + (size_t)rd.cig_off32[ri] + 3 * (size_t)ri; cg.lds = nullptr;
                    } else {
                        cg.glb = nullptr; cg.lds = ((rinfo & I_INB) ? wl.cigB : wl.cigA) + r;
                    }
                    int k = 0;
                    int32_t qq = 0, rr = (int32_t)st[S_POS * TILE + r], oplen = 0;
                    uint32_t op = OP_H;
                    // find the op containing query index j0
                    int32_t opq = 0;   // query index at the start of the current op
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
                        int32_t left = opq + oplen - q;   // bases left in this op
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
                if (bad) atomicOr(&st[S_INFO * TILE + r], I_ERRFLAG);
            }
            wave_sync();
        }

        // =================================== P5: lane = read ===================================
        if (valid) {
            int status = ts.err;
            if (!status && P.do_count) {
                const bool flagged = (st[S_INFO * TILE + lane] & I_ERRFLAG) != 0;
                if (cerr || flagged) {
                    if (!nochunk) {
                        // regular read with an error somewhere: the exact walk decides which comes first
                        NullSink ns;
                        if (use_glb) status = count_read_walk(P, gcur, ts.n, ts.pos, lseq, rd.seq, boff, qual, have_qual, ns);
                        else status = count_read_walk(P, lcur, ts.n, ts.pos, lseq, rd.seq, boff, qual, have_qual, ns);
                    } else {
                        status = cerr;
                    }
                }
            }
            if (out.status) out.status[i] = (uint8_t)status;
            if (status) ++n_err;
        }
        wave_sync();
    }

    // ---- final flush ------------------------------------------------------------------------
    __syncthreads();
    if (win_base != NO_WINDOW) {
        for (int i = tid; i < AMP_NSYM * T_W; i += T_WAVES * 64) {
            uint32_t v = L.win[i];
            if (v) {
                int sym = i / T_W, d = i - sym * T_W;
                atomicAdd(&counts[(size_t)(win_base + d) * AMP_NSYM + sym], v);
            }
        }
    }
    if (n_err) atomicAdd(&ctr[2], n_err);
}

static inline int tile_launch(const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out,
                              uint32_t *gscratch, uint32_t *counts, amp_ins_event *ev, unsigned long long *ctr,
                              long long ev_cap, int n_cu, hipStream_t stream) {
    const int64_t n_tiles = (rd.n_reads + TILE - 1) / TILE;
    if (n_tiles == 0) return 0;
    int64_t max_blocks = (int64_t)n_cu * 2;
    int64_t tpb = (n_tiles + max_blocks - 1) / max_blocks;
    tpb = ((tpb + T_WAVES - 1) / T_WAVES) * T_WAVES;   // whole super-tiles per block
    int64_t grid = (n_tiles + tpb - 1) / tpb;
    k_tile<<<(unsigned)grid, T_WAVES * 64, 0, stream>>>(P, rd, read_base, out, gscratch, counts, ev, ctr, ev_cap, (int)tpb);
    return (int)hipGetLastError();
}

}  // namespace amp
