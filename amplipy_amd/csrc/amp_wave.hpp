// amp_wave.hpp -- one read per WAVE: the second pass's treatment of reads with tens or hundreds of CIGAR ops
// (Nanopore-like amplicon reads), written for CDNA4 / gfx950.
//
// The serial code of amp_read.hpp walks the ops of a read several times per trim stage; on one lane, with the CIGAR
// in global memory, that is tens of dependent memory round trips per stage.  Here the CIGAR lives in the wave's own
// LDS rows and lane = op: every loop of the reference over the ops (A:389-412 get_pos_on_query, A:363-386
// get_pos_on_ref, the clip loops A:467-510 / A:524-555 / A:597-622 / A:658-683, fix_cigar A:415-423) becomes a
// prefix sum over the 64 lanes plus a compaction; the sliding quality window (A:566-587, A:630-649) is evaluated
// at 512 positions at once (lane = 8 consecutive bases), and update_base_counts (A:690-753) runs as lane = 8 bases
// for the match ops and lane = op for deletions and insertion runs.
//
// The path takes reads it can treat exactly with these closed forms: ops M I D N S = X only, no zero-length op,
// query lengths summing to l_seq, qualities present, window <= 8.  wave_read() returns false for anything else
// and the caller runs the serial code (so does a counting error: the exact walk then finds the first one in pair order).
#pragma once

#include "amp_tile.hpp"

namespace amp {

constexpr int WV_MAXOPS = 512;     // words per CIGAR row in LDS (input ops <= WV_MAXOPS - 4)
constexpr int WV_EVCAP = 128;      // insertion events staged per wave

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts: no LDS traffic)
__device__ __forceinline__ uint32_t wv_scan(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return (uint32_t)v;
}
// inclusive prefix maximum (unsigned; 0 is the identity)
__device__ __forceinline__ uint32_t wv_scan_max(uint32_t x) {
    uint32_t v = x, t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = v > t ? v : t;
    return v;
}
__device__ __forceinline__ uint32_t wv_last(uint32_t incl) { return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); }
__device__ __forceinline__ uint32_t wv_sum(uint32_t x) { return wv_last(wv_scan(x)); }

struct WvNullSink {   // dry run of the exact walk: only the status matters
    __device__ void add(int32_t, uint32_t) {}
    __device__ void event(int32_t, int32_t, int32_t) {}
};

// a CIGAR row in LDS seen by the serial code (fallbacks run on lane 0)
struct LdsRow {
    lds_u32 *p;
    __device__ __forceinline__ uint32_t get(int i) const { return p[i]; }
    __device__ __forceinline__ void set(int i, uint32_t v) const { p[i] = v; }
};

// A:389-412 get_pos_on_query: the first reference-consuming op that reaches ref_pos decides.
__device__ __forceinline__ int32_t wv_pos_on_query(const lds_u32 *c, int n, int32_t ref_pos, int32_t ref_start, int lane) {
    int32_t qc = 0, rc = ref_start;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < n;
        const uint32_t v = valid ? c[i] : 0u, op = v & 15u;
        const int32_t len = (int32_t)(v >> 4);
        const int32_t lq = valid && consumes_query(op) ? len : 0, lr = valid && consumes_ref(op) ? len : 0;
        const int32_t iq = (int32_t)wv_scan((uint32_t)lq), ir = (int32_t)wv_scan((uint32_t)lr);
        const int32_t Q = qc + iq - lq, R = rc + ir - lr;
        const unsigned long long hit = __ballot(valid && consumes_ref(op) && ref_pos <= R + len);
        if (hit) {
            const int h = __ffsll((long long)hit) - 1;
            const int32_t res = consumes_query(op) ? Q + (ref_pos - R) : Q;
            return __builtin_amdgcn_readlane(res, h);
        }
        qc += (int32_t)wv_last((uint32_t)iq); rc += (int32_t)wv_last((uint32_t)ir);
    }
    return qc;
}

// A:363-386 get_pos_on_ref
__device__ __forceinline__ int32_t wv_pos_on_ref(const lds_u32 *c, int n, int32_t query_pos, int32_t ref_start, int lane) {
    int32_t qc = 0, rc = ref_start;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < n;
        const uint32_t v = valid ? c[i] : 0u, op = v & 15u;
        const int32_t len = (int32_t)(v >> 4);
        const int32_t lq = valid && consumes_query(op) ? len : 0, lr = valid && consumes_ref(op) ? len : 0;
        const int32_t iq = (int32_t)wv_scan((uint32_t)lq), ir = (int32_t)wv_scan((uint32_t)lr);
        const int32_t Q = qc + iq - lq, R = rc + ir - lr;
        const unsigned long long hit = __ballot(valid && consumes_query(op) && query_pos <= Q + len);
        if (hit) {
            const int h = __ffsll((long long)hit) - 1;
            const int32_t res = consumes_ref(op) ? R + (query_pos - Q) : R;
            return __builtin_amdgcn_readlane(res, h);
        }
        qc += (int32_t)wv_last((uint32_t)iq); rc += (int32_t)wv_last((uint32_t)ir);
    }
    return rc;
}

// appends every lane's 0..2 words to dst in lane order
__device__ __forceinline__ void wv_emit(lds_u32 *dst, int &m, int cnt, uint32_t w0, uint32_t w1) {
    const uint32_t ic = wv_scan((uint32_t)cnt);
    const int at = m + (int)ic - cnt;
    if (cnt >= 1) dst[at] = w0;
    if (cnt == 2) dst[at + 1] = w1;
    m += (int)wv_last(ic);
}

// Primer clip (the per-op rules shared by A:467-510 and A:524-555) of `del` query bases from the front of the walk
// order (rev: the ops are walked from the back, as the reference walks reversed(cigar)).  With Q = query bases in front of
// an op, the loop's state at that op is del_before = max(del - Q, 0); the op that sets pos_start is the first one
// that consumes both and is not used up (del_before < len).  In front of it: query-consuming ops become soft clips (a
// partly used I / S splits), D / N are dropped and advance the start; from it on everything is copied.
// Unmerged result -> dst (walk order), returns its length; adv = reference advance (A:514).
__device__ __forceinline__ int wv_primer_clip(const lds_u32 *src, int n, bool rev, int32_t del, lds_u32 *dst, int lane, int32_t &adv) {
    int32_t qc = 0, sp = 0;
    int m = 0;
    bool resolved = false;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < n;
        const uint32_t v = valid ? src[rev ? n - 1 - i : i] : 0u, op = v & 15u;
        const int32_t len = (int32_t)(v >> 4);
        uint32_t w0 = v, w1 = 0u;
        int cnt = valid ? 1 : 0;
        int32_t add = 0;
        if (!resolved) {
            const int32_t lq = valid && consumes_query(op) ? len : 0;
            const int32_t iq = (int32_t)wv_scan((uint32_t)lq);
            const int32_t Q = qc + iq - lq;
            const int32_t db = del - Q > 0 ? del - Q : 0;
            const bool both = consumes_query(op) && consumes_ref(op);
            const unsigned long long pm = __ballot(valid && both && db < len);
            const int p = pm ? __ffsll((long long)pm) - 1 : 64;
            if (valid && lane <= p) {
                if (lane == p) {
                    if (db > 0) { w0 = ((uint32_t)db << 4) | OP_S; w1 = ((uint32_t)(len - db) << 4) | op; cnt = 2; add = db; }
                } else if (consumes_query(op)) {
                    if (db >= len) { w0 = ((uint32_t)len << 4) | OP_S; if (consumes_ref(op)) add = len; }
                    else if (db > 0) { w0 = ((uint32_t)db << 4) | OP_S; w1 = ((uint32_t)(len - db) << 4) | op; cnt = 2; }
                    else w0 = ((uint32_t)len << 4) | OP_S;
                } else {
                    cnt = 0; add = len;            // D / N (the only ops left on this path): dropped, the start moves on
                }
            }
            resolved = pm != 0ull;
            qc += (int32_t)wv_last((uint32_t)iq);
            sp += (int32_t)wv_sum((uint32_t)add);
        }
        wv_emit(dst, m, cnt, w0, w1);
    }
    adv = sp;
    return m;
}

// Quality clip (A:597-622; A:658-683 over the reversed CIGAR) of `del` bases: soft clips pass through without using
// any of it, M / I / = / X are clipped (the partly used one splits), D / N in front of the cut are dropped; once
// del is used up everything is copied.
__device__ __forceinline__ int wv_quality_clip(const lds_u32 *src, int n, bool rev, int32_t del, lds_u32 *dst, int lane) {
    int32_t qc = 0;
    int m = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < n;
        const uint32_t v = valid ? src[rev ? n - 1 - i : i] : 0u, op = v & 15u;
        const int32_t len = (int32_t)(v >> 4);
        const bool eats = valid && consumes_query(op) && op != OP_S;
        const int32_t lq = eats ? len : 0;
        const int32_t iq = (int32_t)wv_scan((uint32_t)lq);
        const int32_t Q = qc + iq - lq;
        const int32_t db = del - Q > 0 ? del - Q : 0;
        uint32_t w0 = v, w1 = 0u;
        int cnt = valid ? 1 : 0;
        if (valid && db > 0 && op != OP_S) {
            if (eats) {
                w0 = ((uint32_t)(db < len ? db : len) << 4) | OP_S;
                if (len > db) { w1 = ((uint32_t)(len - db) << 4) | op; cnt = 2; }
            } else {
                cnt = 0;
            }
        }
        qc += (int32_t)wv_last((uint32_t)iq);
        wv_emit(dst, m, cnt, w0, w1);
    }
    return m;
}

// fix_cigar (A:415-423): runs of the same op collapse into one.  u[0..m) is in walk order; the merged CIGAR goes to
// dst in CIGAR order (rev: back to front).  z is scratch (run-end prefix sums); the front of u is reused for the ops.
__device__ __forceinline__ int wv_merge(lds_u32 *u, int m, lds_u32 *z, lds_u32 *dst, bool rev, int lane) {
    uint32_t pc = 0;
    int kc = 0;
    for (int c0 = 0; c0 < m; c0 += 64) {
        const int j = c0 + lane;
        const bool valid = j < m;
        const uint32_t v = valid ? u[j] : 0u, op = v & 15u;
        const uint32_t nxt = j + 1 < m ? u[j + 1] & 15u : 16u;
        const bool last = valid && op != nxt;
        const uint32_t ip = wv_scan(valid ? v >> 4 : 0u), il = wv_scan(last ? 1u : 0u);
        wave_sync();                      // every lane holds its words: the compacted ops may overwrite the front of u
        if (last) { const int k = kc + (int)il - 1; z[k] = pc + ip; u[k] = op; }
        pc += wv_last(ip); kc += (int)wv_last(il);
    }
    wave_sync();
    for (int k0 = 0; k0 < kc; k0 += 64) {
        const int k = k0 + lane;
        if (k < kc) dst[rev ? kc - 1 - k : k] = ((z[k] - (k ? z[k - 1] : 0u)) << 4) | u[k];
    }
    wave_sync();
    return kc;
}

constexpr int WV_QSTASH = 520;     // quality bytes of a read kept in the wave's LDS stash (512 + the 8 a forward window may reach into)
// the wave's stash: [0, 528) qualities, [528, 784) the packed bases of the first 512, [784, 1296) 256 16-bit op marks of the base lanes
constexpr int WV_STASH_WORDS = 324;
__device__ __forceinline__ uint2 wv_lds8(const lds_u8 *p) { const amp_u32x2 v = *(const lds_u32x2 *)p; return make_uint2(v.x, v.y); }

// 8 bytes starting at byte B (0..8) of the 16-byte group (a, b)
template <int B>
__device__ __forceinline__ uint2 wv_bytes8(const uint2 a, const uint2 b) {
    const uint32_t w[5] = {a.x, a.y, b.x, b.y, 0u};
    constexpr int d = B >> 2, sh = (B & 3) * 8;
    if (sh == 0) return make_uint2(w[d], w[d + 1]);
    return make_uint2(__builtin_amdgcn_alignbit(w[d + 1], w[d], sh), __builtin_amdgcn_alignbit(w[d + 2], w[d + 1], sh));
}
__device__ __forceinline__ uint32_t wv_sum8(const uint2 x) { return __builtin_amdgcn_sad_u8(x.y, 0u, __builtin_amdgcn_sad_u8(x.x, 0u, 0u)); }

// The sliding-window scan of A:566-587 (reverse strand: returns i = number of leading bases to clip) and A:630-649
// (forward strand: i = number of leading bases kept) over the qualities q[lo .. lo+qlen) of the read: at step i the
// window holds w = min(width, bases left) bases and the scan stops at the first i whose window sums to less than
// min_quality * w.  Every lane tests the 8 steps that belong to its 8 bases; the first failing one in scan order wins.
// `q` is the read's first quality byte (8-byte aligned, readable up to 16 bytes past the read).
// The first WV_QSTASH bytes are also in the wave's LDS stash `wq` (filled when the read was loaded).
__device__ __forceinline__ int32_t wv_quality_scan(const uint8_t *q, const lds_u8 *wq, int32_t lseq, int32_t lo, int32_t qlen, int32_t width,
                                                   int32_t min_quality, bool reverse, int lane) {
    const int32_t W = width < qlen ? width : qlen;
    if (qlen <= 0) return 0;
    const int32_t hi = lo + qlen;
    const int b_first = lo >> 9, b_last = (hi - 1) >> 9;
    for (int blk = reverse ? b_last : b_first; reverse ? blk >= b_first : blk <= b_last; blk += reverse ? -1 : 1) {
        const int32_t p8 = blk * 512 + lane * 8;          // this lane's first base
        // forward: the window of a step starts on its base and needs the 7 bytes after the lane's own 8;
        // reverse: it ends on its base and needs the 7 bytes before
        const int32_t o0 = reverse ? p8 - 8 : p8;
        uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
        if (o0 >= 0 && o0 < lseq) a = o0 < WV_QSTASH ? wv_lds8(wq + o0) : *(const uint2 *)(q + o0);
        if (o0 + 8 >= 0 && o0 + 8 < lseq) b = o0 + 8 < WV_QSTASH ? wv_lds8(wq + o0 + 8) : *(const uint2 *)(q + o0 + 8);
        uint32_t fail = 0u;
#define WV_STEP(B)                                                                                                        \
        {                                                                                                                     \
            const int32_t p = p8 + B;                                                                                         \
            if (reverse) {                                                                                                    \
                const int32_t i = p + 1 - lo;                  /* the step whose window ends on base p */                     \
                if (i >= 1 && i <= qlen) {                                                                                    \
                    const int32_t w = W < i ? W : i;                                                                          \
                    uint2 x = wv_bytes8<B + 1>(a, b);          /* the 8 bytes that end on p */                                \
                    const int drop = (8 - w) * 8;              /* keep the top w of them */                                   \
                    if (drop >= 32) { x.x = 0u; x.y = (x.y >> (drop - 32)) << (drop - 32); }                                  \
                    else if (drop) x.x = (x.x >> drop) << drop;                                                               \
                    if ((int32_t)wv_sum8(x) < min_quality * w) fail |= 1u << B;                                               \
                }                                                                                                             \
            } else {                                                                                                          \
                const int32_t i = p - lo;                                                                                     \
                if (i >= 0 && i < qlen) {                                                                                     \
                    const int32_t w = W < qlen - i ? W : qlen - i;                                                            \
                    uint2 x = wv_bytes8<B>(a, b);              /* the 8 bytes that start on p */                              \
                    const int drop = (8 - w) * 8;              /* keep the low w of them */                                   \
                    if (drop >= 32) { x.y = 0u; x.x = (x.x << (drop - 32)) >> (drop - 32); }                                  \
                    else if (drop) x.y = (x.y << drop) >> drop;                                                               \
                    if ((int32_t)wv_sum8(x) < min_quality * w) fail |= 1u << B;                                               \
                }                                                                                                             \
            }                                                                                                                 \
        }
        WV_STEP(0) WV_STEP(1) WV_STEP(2) WV_STEP(3) WV_STEP(4) WV_STEP(5) WV_STEP(6) WV_STEP(7)
#undef WV_STEP
        const unsigned long long any = __ballot(fail != 0u);
        if (any) {
            if (reverse) {
                const int l = 63 - __clzll((long long)any);
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)fail, l);
                return blk * 512 + l * 8 + (31 - __clz((int)f)) + 1 - lo;
            }
            const int l = __ffsll((long long)any) - 1;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)fail, l);
            return blk * 512 + l * 8 + (__ffs((int)f) - 1) - lo;
        }
    }
    return reverse ? 0 : qlen;
}

// Counters and insertion events of the wave path: counts through the block's LDS window like WinSink; events staged in the
// wave's own area and moved to the list with one reservation per flush (wv_flush_events).
struct WaveSink {
    lds_u32 *win;
    int32_t base;
    uint32_t win_n;       // positions per plane of the window
    uint32_t pitch;       // words per plane
    uint32_t skew;        // 6: position d sits at word d + (d >> 6) of its plane; 31: no skew.  The 8-base lanes of a wave add at
                          // positions 8 apart: without the skew lanes l and l + 8 (64 positions apart) share a bank, an 8-way conflict
    uint32_t ev_plane;    // plane index of the insertion-event tally
    uint32_t *counts;
    const EventBuf &eb;
    uint32_t read;
    lds_u32 *ev, *nev;
    uint32_t ev_cap;      // events the wave's stage holds
    __device__ void add(int32_t r, uint32_t col) {
        const uint32_t d = (uint32_t)(r - base);
        if (d < win_n) lds_add(win + col * pitch + d + (d >> skew), 1u);
        else atomicAdd(&counts[(size_t)r * AMP_NSYM + col], 1u);
    }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
        const uint32_t k = __hip_atomic_fetch_add(nev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k < ev_cap) {
            ev[k * 4] = (uint32_t)pos; ev[k * 4 + 1] = read; ev[k * 4 + 2] = (uint32_t)lo; ev[k * 4 + 3] = (uint32_t)hi;
            const uint32_t d = (uint32_t)(pos - base);
            if (d < win_n) lds_add(win + ev_plane * pitch + d + (d >> skew), 1u);
            else atomicAdd(&eb.ins_at[pos], 1u);
        } else {
            eb.record(pos, read, lo, hi);
        }
    }
};

// the wave's staged events -> the block's shard of the list (one reservation)
__device__ __forceinline__ void wv_flush_events(const EventBuf &eb, lds_u32 *ev, lds_u32 *nev, uint32_t ev_cap, int lane) {
    wave_sync();
    const uint32_t staged = *nev;
    const uint32_t n = staged < ev_cap ? staged : ev_cap;
    if (n == 0u) return;
    const unsigned shard = blockIdx.x & (EV_SHARDS - 1);
    unsigned long long b0 = 0ull;
    if (lane == 0) b0 = atomicAdd(&eb.ctr[16 + shard], (unsigned long long)n);
    b0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
    for (uint32_t k = (uint32_t)lane; k < n; k += 64u)
        if ((long long)(b0 + k) < eb.cap)
            eb.ev[(size_t)shard * (size_t)eb.cap + b0 + k] = amp_ins_event{(int32_t)ev[k * 4], ev[k * 4 + 1], (int32_t)ev[k * 4 + 2], (int32_t)ev[k * 4 + 3]};
    wave_sync();
    if (lane == 0) *nev = 0u;
    wave_sync();
}

// The same for a wave that owns a GRANULE of `cap` list slots reserved ahead of time (k_long): the staged events go into it
// at once, the rest of the granule is marked unused (ref_pos = -1, dropped when the list is read out) and the next
// granule is asked for -- its first slot is still on its way from the atomic when this returns and is only looked at by
// the next flush, so the wave never waits for the list cursor (a flush with a returning atomic cost the wave a few
// microseconds; halving their number took the many-op pass from 1.26 to 0.85 ms).  gran: lane 0 holds the granule's
// first slot.  more = false: the last flush of the wave, no further granule.
__device__ __forceinline__ void wv_flush_events_granule(const EventBuf &eb, lds_u32 *ev, lds_u32 *nev, uint32_t cap, int lane,
                                                        unsigned long long &gran, bool more) {
    wave_sync();
    const uint32_t staged = *nev;
    const uint32_t n = staged < cap ? staged : cap;
    const unsigned shard = blockIdx.x & (EV_SHARDS - 1);
    const unsigned long long b0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(gran >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)gran);
    for (uint32_t k = (uint32_t)lane; k < cap; k += 64u)
        if ((long long)(b0 + k) < eb.cap)
            eb.ev[(size_t)shard * (size_t)eb.cap + b0 + k] =
                k < n ? amp_ins_event{(int32_t)ev[k * 4], ev[k * 4 + 1], (int32_t)ev[k * 4 + 2], (int32_t)ev[k * 4 + 3]} : amp_ins_event{-1, 0u, 0, 0};
    wave_sync();
    if (lane == 0) *nev = 0u;
    wave_sync();
    if (more && lane == 0) gran = atomicAdd(&eb.ctr[16 + shard], (unsigned long long)cap);
}

// One read on one wave: trims (A:426-687), outputs, counts (A:690-753).  x, y, z: the wave's three LDS rows of WV_MAXOPS
// words.  Returns false -- with nothing written anywhere -- when the read is not one for this path.
// The per-read fields, one per lane of a single load instruction (a wave that walks a list issues the load for its
// next read before it starts on the current one).
__device__ __forceinline__ uint32_t wv_header_load(const amp_dev_reads &rd, int64_t i, int lane) {
    uint32_t h = 0u;
    if (lane == 0) h = rd.cig_off32[i];
    else if (lane == 1) h = rd.cig_off32[i + 1];
    else if (lane == 2) h = rd.lseq[i];
    else if (lane == 3) h = (uint32_t)rd.pos[i];
    else if (lane == 4) h = rd.flag[i];
    else if (lane == 5) h = (uint32_t)rd.tlen[i];
    else if (lane == 6) h = rd.seq_off8[i];
    return h;
}
__device__ __forceinline__ uint32_t wv_field(uint32_t h, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)h, k); }

#ifdef AMP_WV_STAMPS
__device__ unsigned long long g_wv_acc[8];
#define WV_T(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); wv_tacc[k] += tn_ - tprev_; tprev_ = tn_; } while (0)
#else
#define WV_T(k) do { } while (0)
#endif
template <class Sink>
__device__ bool wave_read(const KParams &P, const amp_dev_reads &rd, int64_t i, uint32_t hdr, const DevOut &out, Sink &sink, const EventBuf &eb,
                          lds_u32 *x, lds_u32 *y, lds_u32 *z, lds_u8 *wq, int max_ops, int lane, unsigned long long *wv_tacc = nullptr) {
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime(); (void)tprev_;
    const uint32_t c0 = wv_field(hdr, 0);
    int n = (int)(wv_field(hdr, 1) - c0);
    const int32_t lseq = (int32_t)wv_field(hdr, 2);
    int32_t pos = (int32_t)wv_field(hdr, 3);
    const uint32_t flag = wv_field(hdr, 4);
    const int64_t boff = (int64_t)wv_field(hdr, 6) * 8;
    const uint8_t *qual = rd.qual + boff;
    if (n < 1 || n > max_ops || lseq <= 0 || pos < 0 || P.window < 1 || P.window > 8) return false;
    // ---- everything the read needs from memory in one round trip: its ops (row x), the first 512 qualities (stash wq)
    // and packed bases (kept in a register for the counting), the left primer table entry
    uint32_t cw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cw[c] = c * 64 + lane < n ? rd.cig[c0 + c * 64 + lane] : (1u << 4);
    const int32_t q8l = lane * 8;
    uint2 qq0 = make_uint2(0u, 0u), qtail = make_uint2(0u, 0u);
    uint32_t sw0 = 0u;
    if (q8l < lseq) { qq0 = *(const uint2 *)(qual + q8l); sw0 = *(const uint32_t *)(rd.seq + ((boff + q8l) >> 1)); }
    if (lane == 0 && lseq > 512) qtail = *(const uint2 *)(qual + 512);
    int32_t left_max_end = -1;
    if (P.do_trim && (uint32_t)pos < (uint32_t)P.ref_len) left_max_end = P.max_end[pos];
    uint32_t qsum = 0u, rsum = 0u;
    bool dirty = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = c * 64 + lane;
        const uint32_t v = cw[c], op = v & 15u;
        if (k < n) x[k] = v;
        dirty |= !((0x19Fu >> op) & 1u) || (v >> 4) == 0u;         // M I D N S = X, no empty op
        qsum += consumes_query(op) && k < n ? v >> 4 : 0u;
        rsum += consumes_ref(op) && k < n ? v >> 4 : 0u;
    }
    for (int c = 256; c < n; c += 64) {
        const int k = c + lane;
        const uint32_t v = k < n ? rd.cig[c0 + k] : (1u << 4), op = v & 15u;
        if (k < n) x[k] = v;
        dirty |= !((0x19Fu >> op) & 1u) || (v >> 4) == 0u;
        qsum += consumes_query(op) && k < n ? v >> 4 : 0u;
        rsum += consumes_ref(op) && k < n ? v >> 4 : 0u;
    }
    *(lds_u32x2 *)(wq + q8l) = amp_u32x2{qq0.x, qq0.y};
    *(lds_u32 *)(wq + 528 + lane * 4) = sw0;
    if (lane == 0) *(lds_u32x2 *)(wq + 512) = amp_u32x2{qtail.x, qtail.y};
    qsum = wv_sum(qsum); rsum = wv_sum(rsum);
    if (__ballot(dirty) || qsum != (uint32_t)lseq || ((uint32_t)__builtin_amdgcn_readfirstlane((int)qq0.x) & 0xFFu) == 0xFFu) return false;   // QUAL '*'
    wave_sync();
    WV_T(0);
    uint32_t tflags = 0u;
    if (P.do_trim) {
        const bool is_paired = flag & 1u, is_reverse = (flag & 0x10u) != 0;
        const int32_t re1 = pos + (int32_t)(rsum ? rsum : 1u) - 1;
        if ((uint32_t)pos >= (uint32_t)P.ref_len || (uint32_t)re1 >= (uint32_t)P.ref_len) return false;     // A:450-451 raise
        const int32_t right_min_start = P.min_start[re1];
        const int32_t tl = (int32_t)wv_field(hdr, 5), at = tl < 0 ? -tl : tl;
        const bool isize_flag = ((int64_t)at - P.max_primer_len) > (int64_t)lseq;                            // A:452
        const bool do_left = !(is_paired && isize_flag && is_reverse) && left_max_end >= 0;                  // A:460
        const bool do_right = !(is_paired && isize_flag && !is_reverse) && right_min_start >= 0;             // A:517
        // the clips below change x: a case left to the serial code has to be seen before the first of them
        int32_t del_l = 0;
        if (do_left) {
            del_l = wv_pos_on_query(x, n, left_max_end + 1, pos, lane);                                      // A:463
            if (del_l < 0) return false;
        }
        if (do_left) {
            tflags |= AMP_TRIM_PRIMER_START;
            int32_t adv;
            const int m = wv_primer_clip(x, n, false, del_l, y, lane, adv);
            n = wv_merge(y, m, z, x, false, lane);
            pos += adv;                                                                                      // A:514
        }
        if (do_right) {
            tflags |= AMP_TRIM_PRIMER_END;
            const int32_t del = lseq - wv_pos_on_query(x, n, right_min_start, pos, lane);                    // A:520
            int32_t adv;
            const int m = wv_primer_clip(x, n, true, del, y, lane, adv);
            n = wv_merge(y, m, z, x, true, lane);
        }
        WV_T(1);
        // A:561 query_alignment_qualities of the clipped read: leading soft clips, trailing soft clips (element 0 is never
        // looked at from the back)
        int32_t qs = 0, trail = 0;
        {
            int first_other = n, last_other = -1;
            for (int c = 0; c < n; c += 64) {
                const int k = c + lane;
                const unsigned long long o = __ballot(k < n && (x[k] & 15u) != OP_S);
                if (o) { if (first_other == n) first_other = c + __ffsll((long long)o) - 1; last_other = c + 63 - __clzll((long long)o); }
            }
            uint32_t a = 0u, b = 0u;
            for (int c = 0; c < n; c += 64) {
                const int k = c + lane;
                const uint32_t len = k < n ? x[k] >> 4 : 0u;
                a += k < first_other ? len : 0u;
                b += k > last_other && k >= 1 ? len : 0u;
            }
            qs = (int32_t)wv_sum(a); trail = (int32_t)wv_sum(b);
        }
        int32_t lo, hi;
        py_slice(qs, lseq - trail, lseq, lo, hi);
        const int32_t qlen = hi - lo;
        const int32_t isc = wv_quality_scan(qual, wq, lseq, lo, qlen, P.window, P.min_quality, is_reverse, lane);
        WV_T(2);
        if (is_reverse) {
            const int32_t del = isc;
            const int32_t sp = wv_pos_on_ref(x, n, del + qs - 1, pos, lane);                                 // A:591
            if (sp > pos) {                                                                                  // A:594
                tflags |= AMP_TRIM_QUALITY;
                const int m = wv_quality_clip(x, n, false, del, y, lane);
                n = wv_merge(y, m, z, x, false, lane);                      // reference_start is NOT advanced
            }
        } else {
            const int32_t del = qlen - isc;
            if (del != 0) {                                                                                  // A:656
                tflags |= AMP_TRIM_QUALITY;
                const int m = wv_quality_clip(x, n, true, del, y, lane);
                n = wv_merge(y, m, z, x, true, lane);
            }
        }
    }
    WV_T(3);
    // ---- the final CIGAR: query / reference start of every op (rows y, z), shape
    int32_t qtot = 0, rtot = 0;
    int first_body = -1, last_body = -1, n_body = 0;
    bool twin_ins = false;
    {
        uint32_t *home = out.new_cig + (size_t)c0 + 3 * (size_t)i;
        for (int c = 0; c < n; c += 64) {
            const int k = c + lane;
            const bool valid = k < n;
            const uint32_t v = valid ? x[k] : 0u, op = v & 15u;
            const int32_t len = (int32_t)(v >> 4);
            if (valid) home[k] = v;
            const int32_t lq = valid && consumes_query(op) ? len : 0, lr = valid && consumes_ref(op) ? len : 0;
            const int32_t iq = (int32_t)wv_scan((uint32_t)lq), ir = (int32_t)wv_scan((uint32_t)lr);
            if (valid) { y[k] = (uint32_t)(qtot + iq - lq); z[k] = (uint32_t)(pos + rtot + ir - lr); }
            const unsigned long long body = __ballot(valid && op != OP_S);
            if (body) {
                if (first_body < 0) first_body = c + __ffsll((long long)body) - 1;
                last_body = c + 63 - __clzll((long long)body);
                n_body += __popcll(body);
            }
            twin_ins |= __ballot(valid && op == OP_I && k + 1 < n && (x[k + 1] & 15u) == OP_I) != 0ull;
            qtot += (int32_t)wv_last((uint32_t)iq); rtot += (int32_t)wv_last((uint32_t)ir);
        }
    }
    wave_sync();
    const int32_t ref_len_final = rtot ? rtot : 1;
    if (lane == 0) {
        if (out.new_pos) out.new_pos[i] = pos;
        if (out.new_ncig) out.new_ncig[i] = (uint32_t)n;
        if (out.ref_len) out.ref_len[i] = ref_len_final;
        if (out.trim_flags) out.trim_flags[i] = (uint8_t)tflags;
    }
    WV_T(4);
    int err = 0;
    if (P.do_count) {
        const uint32_t G = (uint32_t)P.ref_len;
        const int32_t mq = P.min_quality;
        const bool regular = (n_body == 0 || last_body - first_body + 1 == n_body) && !twin_ins;
        if (!regular) {
            // soft clips inside the alignment, two insertions in a row: the exact walk on one lane, CIGAR in LDS
            if (lane == 0) err = count_read_walk(P, LdsRow{x}, n, pos, lseq, ReadBytesCached{rd.seq, boff, qual}, true, sink);
        } else if (n_body) {
            const int32_t ref_end = pos + ref_len_final;
            bool bad = false;
            // match bases: lane = base, 256 consecutive query bases a turn (four per lane, 64 apart).  The ops that start inside
            // the turn's range mark their first base with their index; a running maximum over the bases gives every base its op
            // (the op that reaches in from the bases before is carried over).  All LDS reads of a step are issued together:
            // the turn is a handful of dependent LDS round trips, not one chain per 64 bases.
            {
                lds_u16 *const mark = (lds_u16 *)(wq + 784);
                const lds_u8 *const sq = wq + 528;
                int kbase = 0;
                uint32_t carry = 0u;
                for (int32_t Q0 = 0; Q0 < lseq; Q0 += 256) {
                    *(lds_u32x2 *)(mark + lane * 4) = amp_u32x2{0u, 0u};
                    for (;;) {
                        const int k = kbase + lane;
                        const int32_t yk = k < n ? (int32_t)y[k] : INT32_MAX;
                        const bool inr = yk < Q0 + 256;
                        if (inr && consumes_query(x[k] & 15u)) mark[yk - Q0] = (uint16_t)(k + 1);
                        const unsigned long long m = __ballot(inr);
                        if (m == ~0ull) { kbase += 64; continue; }
                        kbase += __popcll(m);                 // (ops are sorted by their query start: the lanes inside form a prefix)
                        break;
                    }
                    uint32_t v[4], qb[4], sb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int32_t q = Q0 + r * 64 + lane;
                        v[r] = mark[r * 64 + lane];
                        qb[r] = 0u; sb[r] = 0u;
                        if (q < lseq) {
                            qb[r] = q < WV_QSTASH ? (uint32_t)wq[q] : (uint32_t)qual[q];
                            sb[r] = q < 512 ? (uint32_t)sq[q >> 1] : (uint32_t)rd.seq[(boff + q) >> 1];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = wv_scan_max(v[r]);
                        v[r] = v[r] > carry ? v[r] : carry;
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)v[r], 63);
                    }
                    uint32_t w[4];
                    int32_t rr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int32_t q = Q0 + r * 64 + lane;
                        const int idx = (int)v[r] - 1;
                        w[r] = 15u; rr[r] = 0;
                        if (q < lseq) { w[r] = x[idx]; rr[r] = (int32_t)z[idx] + (q - (int32_t)y[idx]); }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int32_t q = Q0 + r * 64 + lane;
                        if (q < lseq && is_match_op(w[r] & 15u) && (int32_t)qb[r] >= mq) {                    // A:718
                            const uint32_t col = col_of_code((q & 1) ? (sb[r] & 15u) : (sb[r] >> 4));
                            if (col > 4u || (uint32_t)rr[r] >= G) bad = true;
                            else sink.add(rr[r], col);                                                       // A:751-753
                        }
                    }
                }
            }
            WV_T(5);
            // deletions / reference skips and insertion runs: lane = op
            const QualAt qglob{qual};
            const auto qf = [&](int32_t q) -> uint32_t { return q < WV_QSTASH ? (uint32_t)wq[q] : qglob(q); };
            for (int c = 0; c < n; c += 64) {
                const int k = c + lane;
                if (k >= n) continue;
                const uint32_t v = x[k], op = v & 15u;
                const int32_t len = (int32_t)(v >> 4);
                if (op == OP_D || op == OP_N) {                                                              // A:714-715
                    const int32_t r0 = (int32_t)z[k];
                    for (int32_t j = 0; j < len; ++j) {
                        if ((uint32_t)(r0 + j) >= G) { bad = true; break; }
                        sink.add(r0 + j, 5u);
                    }
                } else if (op == OP_I) {                                                                     // A:730-748
                    // one event per maximal run of good bases.  A run that ends inside the op was stopped by a low base,
                    // which the reference swallows: counted at reference_end - 1.  A run that reaches the op's end looks
                    // at the pair behind it: a match base (anchor; handed back), a deleted base (q is None: the
                    // slice runs to the read's end), the first base of the end clip (stops the scan: reference_end
                    // - 1 again), or nothing (get_aligned_pairs exhausted: IndexError).
                    const int32_t qa = (int32_t)y[k], r2 = (int32_t)z[k];
                    const uint32_t nop = k + 1 < n ? x[k + 1] & 15u : 16u;
                    int32_t j = 0;
                    while (j < len) {
                        if ((int32_t)qf(qa + j) < mq) { ++j; continue; }                                     // A:718
                        const int32_t js = j;
                        while (j < len && (int32_t)qf(qa + j) >= mq) ++j;
                        int32_t slo, shi, ins_pos;
                        if (j < len) {
                            py_slice(qa + js - 1, qa + j, lseq, slo, shi);                                   // A:738
                            ins_pos = ref_end; ++j;                                                          // A:739-740
                        } else if (is_match_op(nop)) {
                            if (r2 == 0) py_slice(qa + js, qa + len + 1, lseq, slo, shi);                    // A:735-736
                            else py_slice(qa + js - 1, qa + len, lseq, slo, shi);
                            ins_pos = r2;                                                                    // A:742
                        } else if (nop == OP_D || nop == OP_N) {
                            if (r2 == 0) { bad = true; break; }                                              // None + 1: TypeError
                            py_slice(qa + js - 1, lseq, lseq, slo, shi);                                     // seq[a:None]
                            ins_pos = r2;
                        } else if (nop == OP_S) {
                            py_slice(qa + js - 1, qa + len, lseq, slo, shi);
                            ins_pos = ref_end;
                        } else {
                            bad = true; break;                                                               // A:734 IndexError
                        }
                        ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;                                         // A:744
                        if ((uint32_t)ins_pos >= G) { bad = true; break; }
                        sink.event(ins_pos, slo, shi);
                    }
                }
            }
            WV_T(6);
            if (__ballot(bad)) {
                // some pair cannot be counted: the exact walk names the first error in pair order
                if (lane == 0) {
                    WvNullSink ns;
                    err = count_read_walk(P, LdsRow{x}, n, pos, lseq, ReadBytesCached{rd.seq, boff, qual}, true, ns);
                }
            }
        }
    }
    if (lane == 0) {
        if (out.status) out.status[i] = (uint8_t)err;
        if (err) atomicAdd(&eb.ctr[2], 1ull);
    }
    wave_sync();
    return true;
}

// ---- the kernel of its own for batches made of such reads (Nanopore-like amplicon runs) -------------------------
// The heavy pass gives a list segment to a block of four waves, two blocks per CU; with every read of a batch on
// this path that is 8 reads in flight per CU, each a chain of dependent steps.  k_long runs 24 waves per CU (two
// blocks of twelve; the trims need 77 registers) over the list of long reads that k_gcompact extracts from the general
// list (entries flagged GL_LONG, which the tile kernel then skips).  A block takes chunks of L_CHUNK consecutive list
// entries from a ticket counter in memory -- neighbours on the reference: their counts meet in one LDS window that is
// flushed per chunk --; its waves draw the chunk's reads from a ticket counter in LDS, each with the header of its next
// read already on the way.
// A read the closed forms do not take loses its flag: the general pass, launched behind this kernel, treats it.
// (Measured on 200,000 reads of 45 ops: one block of 16 waves per CU 0.99 ms for the whole pass, two blocks of 12 waves
// 0.85 ms -- but 1.26 ms with event stages of 32 instead of 64: every flush of a stage is a returning atomic on the list
// cursor that the wave waits for; waves taking every 16th read of the chunk wait 19 % longer at the chunk's barrier than
// with the LDS ticket; a block that walks a contiguous 1/256 of the list with a moving window has fewer barriers but ran
// 0.76-2.2 ms from launch to launch.)
constexpr int L_WAVES = 12;
constexpr int L_MAXOPS = 160;        // words per CIGAR row: reads of up to L_MAXOPS - 4 ops (more: the heavy pass's wave path, 508)
constexpr int L_EVCAP = 64;          // events staged per wave
constexpr int L_CHUNK = 128;         // most list entries a block takes at a time
constexpr uint32_t L_WIN = 1024;     // reference positions of the block's window
constexpr uint32_t L_PITCH = L_WIN + L_WIN / 64;   // words per plane (skewed: see WaveSink)
struct LongLds {
    uint32_t win[(AMP_NSYM + 1) * L_PITCH];
    uint32_t rows[L_WAVES * 3 * L_MAXOPS];
    uint32_t ev[L_WAVES * L_EVCAP * 4];
    uint32_t wq[L_WAVES * WV_STASH_WORDS];
    uint32_t nev[L_WAVES];
    uint32_t chunk, ticket;
    int32_t base;
};

__global__ void __launch_bounds__(L_WAVES * 64, 6)
k_long(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *counts, EventBuf eb,
       const uint32_t *__restrict__ llist, const uint32_t *__restrict__ lpos, uint32_t *dense) {
    __shared__ __attribute__((aligned(16))) LongLds L;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_long = (uint32_t)eb.ctr[26];            // left by k_gcompact
    if (n_long == 0u) return;
    for (uint32_t w = (uint32_t)tid; w < (AMP_NSYM + 1) * L_PITCH; w += L_WAVES * 64) L.win[w] = 0u;
    if (lane == 0) L.nev[wave] = 0u;
    lds_u32 *const row = (lds_u32 *)L.rows + wave * (3 * L_MAXOPS);
    lds_u32 *const wev = (lds_u32 *)L.ev + wave * (L_EVCAP * 4), *const wn = (lds_u32 *)&L.nev[wave];
    lds_u8 *const wq = (lds_u8 *)((lds_u32 *)L.wq + wave * WV_STASH_WORDS);
    // chunk sizes: at least four chunks a block, at most L_CHUNK entries; the last quarter of the list goes out in chunks a
    // quarter that size, so that the blocks still busy at the end are busy with little (with 128 everywhere and three
    // chunks a block, the few blocks that drew a fourth chunk set the kernel's duration)
    uint32_t big = n_long / (4u * gridDim.x);
    big = big > (uint32_t)L_CHUNK ? (uint32_t)L_CHUNK : big < 16u ? 16u : big;
    const uint32_t small = big / 4u < 16u ? 16u : big / 4u;
    const uint32_t n_bigc = (uint32_t)(((unsigned long long)n_long * 3ull / 4ull) / big), rest0 = n_bigc * big;
    const uint32_t n_chunk = n_bigc + (n_long - rest0 + small - 1u) / small;
    unsigned long long gran = 0ull;                          // the wave's granule of the event list (lane 0)
    if (lane == 0) gran = atomicAdd(&eb.ctr[16 + (blockIdx.x & (EV_SHARDS - 1))], (unsigned long long)L_EVCAP);
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const uint32_t c = (uint32_t)atomicAdd(&eb.ctr[27], 1ull);
            L.chunk = c; L.ticket = 0u;
            if (c < n_chunk) { const int32_t p = rd.pos[llist[c < n_bigc ? c * big : rest0 + (c - n_bigc) * small]]; L.base = (p < 0 ? 0 : p) & ~31; }
        }
        __syncthreads();
        const uint32_t ch = L.chunk;
        if (ch >= n_chunk) break;
        const int32_t base = L.base;
        const uint32_t k0 = ch < n_bigc ? ch * big : rest0 + (ch - n_bigc) * small;
        const uint32_t k1e = k0 + (ch < n_bigc ? big : small), k1 = k1e < n_long ? k1e : n_long;
        const auto draw = [&]() -> uint32_t {
            uint32_t t = 0u;
            if (lane == 0) t = __hip_atomic_fetch_add((lds_u32 *)&L.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return k0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        };
        uint32_t k = draw();
        int64_t i = 0;
        uint32_t hdr = 0u;
        if (k < k1) { i = (int64_t)llist[k]; hdr = wv_header_load(rd, i, lane); }
        while (k < k1) {
            const uint32_t kn = draw();
            int64_t in = 0;
            uint32_t hn = 0u;
            if (kn < k1) { in = (int64_t)llist[kn]; hn = wv_header_load(rd, in, lane); }      // lands while this read is worked on
            WaveSink ws{(lds_u32 *)L.win, base, L_WIN, L_PITCH, 6u, (uint32_t)AMP_NSYM, counts, eb, (uint32_t)(read_base + (uint64_t)i), wev, wn, (uint32_t)L_EVCAP};
            if (!wave_read(P, rd, i, hdr, out, ws, eb, row, row + L_MAXOPS, row + 2 * L_MAXOPS, wq, L_MAXOPS - 4, lane, tacc)) {
                if (lane == 0) { dense[lpos[k]] &= ~GL_LONG; atomicAdd(&eb.ctr[28], 1ull); }
            }
            // (a read of this kind records about ten events; one that overflows the stage records the rest one by one)
            if (*wn > (uint32_t)L_EVCAP - 24u) wv_flush_events_granule(eb, wev, wn, (uint32_t)L_EVCAP, lane, gran, true);
            k = kn; i = in; hdr = hn;
        }
        __syncthreads();
        for (uint32_t w = (uint32_t)tid; w < (AMP_NSYM + 1) * L_WIN; w += L_WAVES * 64) {
            const uint32_t plane = w / L_WIN, d = w % L_WIN, at = plane * L_PITCH + d + (d >> 6);
            const uint32_t v = L.win[at];
            if (!v) continue;
            L.win[at] = 0u;
            const uint32_t p = (uint32_t)base + d;
            if (plane < AMP_NSYM) atomicAdd(&counts[(size_t)p * AMP_NSYM + plane], v);
            else atomicAdd(&eb.ins_at[p], v);
        }
    }
    wv_flush_events_granule(eb, wev, wn, (uint32_t)L_EVCAP, lane, gran, false);
#ifdef AMP_WV_STAMPS
    if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&eb.ctr[8 + k], tacc[k]);
#endif
}

}  // namespace amp
