// amplihip.hip -- libamplihip.so: C ABI (include/amplihip.h) + HIP kernels for gfx950.
//
// Replaces AmpliPy.py's per-read loop (A:896-915 = trim_read A:426-687 +
// update_base_counts A:690-753) and the integer part of calling (A:756-771, A:917-952)
// with batch kernels.  There is no host execution path for the read work: every entry point
// that touches reads launches kernels on the ctx's device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <new>
#include <vector>

#include "../../include/amplihip.h"
#include "amp_read.hpp"
#include "amp_tile.hpp"
#include "amp_fast.hpp"
#include "amp_fast5.hpp"
#include "amp_fast6.hpp"
#include "amp_fast7.hpp"
#include "amp_wave.hpp"
#include "amp_ins.hpp"

using namespace amp;

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
struct DBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = std::max(bytes, cap + cap / 2);
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, ncap);
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
        }
        if (p) (void)hipFree(p);
        p = np; cap = ncap;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

struct amp_ctx {
    int device = 0;
    int32_t ref_len = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t *d_counts = nullptr;
    bool own_counts = false;
    int32_t *d_min_start = nullptr, *d_max_end = nullptr;
    int32_t max_primer_len = 0;
    bool have_primers = false;
    int32_t min_quality = 20, window = 4, do_trim = 1, do_count = 1;
    // insertion events
    DBuf events;                  // amp_ins_event[EV_SHARDS][ev_cap]
    int64_t ev_cap = 0;           // per shard
    bool ev_reserved = false;     // caller sized the buffer: skip the bound pre-pass
    unsigned long long *d_ctr = nullptr;  // [0] events recorded, [1] event bound, [2] error reads, [3] deferred reads
    uint32_t *d_ins_at = nullptr;         // [ref_len] insertion events per reference position
    uint8_t *d_ref = nullptr;             // [ref_len] reference sequence, ASCII (amp_set_reference)
    bool have_ref = false;
    // staging for the host-pointer path
    DBuf s_pos, s_flag, s_tlen, s_lseq, s_cigoff, s_cig, s_seqoff, s_seq, s_qual;
    int64_t staged_n = -1, staged_ncig = 0, staged_nbases = 0;     // the batch of the last amp_process_batch, still in the buffers above
    DBuf o_pos, o_ncig, o_cig, o_reflen, o_flags, o_status;
    DBuf scratch;                 // CIGAR scratch for reads whose ops do not fit the LDS slots
    DBuf call_buf;
    DBuf agg_buf;                  // amp_aggregate_ins_events: run records + the sort's scratch
    void *h_pin = nullptr; size_t h_pin_cap = 0;   // pinned staging for call results (amp_call_compact_begin: written by the kernel itself)
    std::vector<uint8_t> h_img; bool img_pinned = false;   // ... and the pageable image of a view that was not begun (one copy)
    bool call_pending = false;     // amp_call_compact_begin has enqueued the calling kernels; amp_call_compact_view picks them up
    amp_call_params call_pending_params{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, ev_call = nullptr;
    bool timed = false;
    bool last_split = false;      // the last launch recorded ev1 / ev2
    int last_kv = 0;              // kernel variant the last batch took (amp_last_kernel_variant)
    bool split_timing = false;    // also time the first kernel of a pass alone (amp_set_timing): costs an idle gap behind it
    int n_cu = 256;
    int cu_share = 1;              // the fast kernels of this ctx are sized for n_cu / cu_share CUs (amp_set_cu_share)
    uint32_t epoch = 0;            // launches of the read kernels so far (KParams::epoch)
    int kernel_variant = 0;       // 0 = by the batch (4 for reads of up to 152 padded bases on average, else 5), 5 = fast kernel (second generation) + general pass, 4 = its first generation, 1 = one lane per read (reference kernels), 2 = fused tile kernel, 3 = k_trim + k_scan + k_tile<SPLIT>,
                                  // 4 = k_fast (simple reads, one pass over their bytes) + k_tile<LIST> over the others
    uint32_t *dbg_dcnt = nullptr; int dbg_grid = 0;
    uint32_t phases = 0xFFu;       // always 0xFF in the shipped library; -DAMP_DEV builds can mask phases of the tile kernel (AMPLIHIP_PHASES)
    char err[320] = {0};
};

static hipError_t grow_events(amp_ctx *c, int64_t ncap);

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,               \
                     hipGetErrorString(e__), __FILE__, __LINE__);                                  \
            return e__ == hipErrorOutOfMemory ? AMP_ENOMEM : AMP_EHIP;                             \
        }                                                                                          \
    } while (0)

static hipError_t grow_events(amp_ctx *c, int64_t ncap) {   // re-lays the shard regions out for a larger capacity
    if (ncap <= c->ev_cap) return hipSuccess;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, (size_t)EV_SHARDS * (size_t)ncap * sizeof(amp_ins_event));
    if (e != hipSuccess) return e;
    if (c->events.p && c->ev_cap) {
        for (int s = 0; s < EV_SHARDS && e == hipSuccess; ++s)
            e = hipMemcpyAsync((amp_ins_event *)np + (size_t)s * ncap, c->events.as<amp_ins_event>() + (size_t)s * c->ev_cap,
                               (size_t)c->ev_cap * sizeof(amp_ins_event), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(np); return e; }
    }
    c->events.release();
    c->events.p = np; c->events.cap = (size_t)EV_SHARDS * (size_t)ncap * sizeof(amp_ins_event);
    c->ev_cap = ncap;
    return hipSuccess;
}

struct Guard {  // make the ctx's device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit Guard(amp_ctx *c) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) ok = hipSetDevice(c->device) == hipSuccess;
    }
    ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---------------------------------------------------------------------------------------
// kernels shared by both variants
// ---------------------------------------------------------------------------------------

// Upper bound on the insertion events a batch can record: every event starts on its own
// (q, None) aligned pair inside [query_alignment_start, query_alignment_end), i.e. on a base
// of an I / P / inner-S op.  Trimming only turns such bases into clips, never creates them.
__global__ void k_event_bound(int64_t n, const uint32_t *__restrict__ cig_off, const uint32_t *__restrict__ cig,
                              unsigned long long *ctr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b = 0;
    if (i < n) {
        uint32_t c0 = cig_off[i], c1 = cig_off[i + 1];
        unsigned long long lead = 0, all = 0, trail = 0;
        bool in_lead = true;
        for (uint32_t k = c0; k < c1; ++k) {
            uint32_t v = cig[k], op = v & 15u, len = v >> 4;
            if (op == OP_H) continue;
            if (op == OP_S) { all += len; trail += len; if (in_lead) lead += len; }
            else { in_lead = false; trail = 0; if (op == OP_I || op == OP_P) all += len; }
        }
        b = all - lead - (in_lead ? 0 : trail);
    }
    for (int o = 32; o > 0; o >>= 1) b += __shfl_down(b, o);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&ctr[1], b);
}

struct DevSink {
    uint32_t *counts;
    const EventBuf &eb;
    uint32_t read;
    __device__ void add(int32_t r, uint32_t col) { atomicAdd(&counts[(size_t)r * AMP_NSYM + col], 1u); }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) { eb.record(pos, read, lo, hi); }
};


// Second-pass sink: the reads of one list segment come from one contiguous range of the sorted batch,
// so their counts are privatised in a block-wide LDS window (device-scope atomics on distinct
// addresses run at ~23 G/s chip-wide on MI355X: they, not the walk, bound a lane-per-read pass).
// Positions outside the window go straight to the global table.
constexpr uint32_t D_WIN = 1024;            // positions of the second pass's LDS window (with 512 the 400-base reads of a segment fell outside it half the time: global atomics)
constexpr uint32_t D_PLANES = AMP_NSYM + 1;   // six symbols + the insertion-event tally
constexpr uint32_t D_EVCAP = 512;             // events staged per round (the rest go out one by one)
struct WinSink {
    lds_u32 *win;
    int32_t base;
    uint32_t *counts;
    const EventBuf &eb;
    uint32_t read;
    lds_u32 *ev;      // [D_EVCAP][4] staged events
    lds_u32 *nev;     // staging cursor (keeps counting past D_EVCAP)
    __device__ void add(int32_t r, uint32_t col) {
        const uint32_t d = (uint32_t)(r - base);
        if (d < D_WIN) lds_add(win + col * D_WIN + d, 1u);
        else atomicAdd(&counts[(size_t)r * AMP_NSYM + col], 1u);
    }
    // One returning atomic per event on the list cursor serialises in L2 (~0.2 events/ns chip-wide,
    // measured): events are staged in LDS and the block reserves list space once per round.
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
        const uint32_t k = atomicAdd((uint32_t *)nev, 1u);
        if (k < D_EVCAP) {
            ev[k * 4] = (uint32_t)pos; ev[k * 4 + 1] = read; ev[k * 4 + 2] = (uint32_t)lo; ev[k * 4 + 3] = (uint32_t)hi;
            const uint32_t d = (uint32_t)(pos - base);
            if (d < D_WIN) lds_add(win + AMP_NSYM * D_WIN + d, 1u);
            else atomicAdd(&eb.ins_at[pos], 1u);
        } else {
            eb.record(pos, read, lo, hi);
        }
    }
};

// Sink of the light second pass: a handful of '-' counts per read go straight to the table, events are staged.
struct StageSink {
    uint32_t *counts;
    const EventBuf &eb;
    uint32_t read;
    lds_u32 *ev, *nev;
    __device__ void add(int32_t r, uint32_t col) { atomicAdd(&counts[(size_t)r * AMP_NSYM + col], 1u); }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
        const uint32_t k = atomicAdd((uint32_t *)nev, 1u);
        if (k < D_EVCAP) {
            ev[k * 4] = (uint32_t)pos; ev[k * 4 + 1] = read; ev[k * 4 + 2] = (uint32_t)lo; ev[k * 4 + 3] = (uint32_t)hi;
            atomicAdd(&eb.ins_at[pos], 1u);
        } else {
            eb.record(pos, read, lo, hi);
        }
    }
};

// Staged events -> the block's shard of the list, one reservation for all of them.  Block-wide.
__device__ void flush_staged_events(const EventBuf &eb, const uint32_t *s_ev, uint32_t *s_nev, unsigned long long *s_evbase) {
    __syncthreads();
    const uint32_t nev = *s_nev < D_EVCAP ? *s_nev : D_EVCAP;
    const unsigned shard = blockIdx.x & (EV_SHARDS - 1);
    if (threadIdx.x == 0 && nev) *s_evbase = atomicAdd(&eb.ctr[16 + shard], (unsigned long long)nev);
    __syncthreads();
    if (nev) {
        const unsigned long long eb0 = *s_evbase;
        for (uint32_t k = threadIdx.x; k < nev; k += blockDim.x)
            if ((long long)(eb0 + k) < eb.cap)
                eb.ev[(size_t)shard * (size_t)eb.cap + eb0 + k] =
                    amp_ins_event{(int32_t)s_ev[k * 4], s_ev[k * 4 + 1], (int32_t)s_ev[k * 4 + 2], (int32_t)s_ev[k * 4 + 3]};
    }
    __syncthreads();
    if (threadIdx.x == 0) *s_nev = 0;
    __syncthreads();
}

// Sink of a read with a long CIGAR (tens of insertions): one reservation on the list cursor for the whole read
// (an upper bound of its events) instead of one per event; slots left over are marked invalid
// (ref_pos = -1) and dropped when the list is read out.  Counts go through the block's window.
struct SliceSink {
    WinSink &w;
    amp_ins_event *slice;
    uint32_t cap, used;
    __device__ void add(int32_t r, uint32_t col) { w.add(r, col); }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
        if (used < cap) {
            slice[used++] = amp_ins_event{pos, w.read, lo, hi};
            const uint32_t d = (uint32_t)(pos - w.base);
            if (d < D_WIN) lds_add(w.win + AMP_NSYM * D_WIN + d, 1u);
            else atomicAdd(&w.eb.ins_at[pos], 1u);
        } else {
            w.event(pos, lo, hi);
        }
    }
};

struct NullSink {   // dry run: only the status matters
    __device__ void add(int32_t, uint32_t) {}
    __device__ void event(int32_t, int32_t, int32_t) {}
};

// one CIGAR column in LDS for the second pass (stride = block size)
struct LdsCig256 {
    lds_u32 *p;
    __device__ __forceinline__ uint32_t get(int i) const { return p[i * 256]; }
    __device__ __forceinline__ void set(int i, uint32_t v) const { p[i * 256] = v; }
};

__device__ __forceinline__ bool is_home(const CigBuf<1> &b, const uint32_t *home) { return b.p == home; }
__device__ __forceinline__ bool is_home(const LdsCig256 &, const uint32_t *) { return false; }

// One read, start to finish, on one lane with the serial code of amp_read.hpp.  cur / tmp are the
// two CIGAR buffers the trims ping-pong between (global slots, or LDS columns when the read fits);
// the final CIGAR always lands in the read's output slot.  status_only: the tile kernel already
// counted this read and only needs to know which error comes first in pair order.
template <class CB, class Sink>
__device__ void process_read_body(const KParams &P, const amp_dev_reads &rd, int64_t i, const DevOut &out, Sink &sink,
                                  const EventBuf &eb, bool status_only, CB cur, CB tmp, uint32_t *home, uint32_t c0, int n) {
    for (int k = 0; k < n; ++k) cur.set(k, rd.cig[c0 + k]);
    const int32_t lseq = (int32_t)rd.lseq[i];
    const int64_t boff = (int64_t)rd.seq_off8[i] * 8;
    const uint8_t *qual = rd.qual + boff;
    const bool have_qual = lseq > 0 && qual[0] != 0xFF;
    TrimState st{rd.pos[i], n, 0u, 0};
    if (P.do_trim) trim_read_serial(P, st, rd.flag[i], rd.tlen[i], lseq, qual, have_qual, cur, tmp);
    if (!st.err && !is_home(cur, home))
        for (int k = 0; k < st.n; ++k) home[k] = cur.get(k);
    int err = st.err;
    if (!err && P.do_count) {
        if (status_only) {
            NullSink ns;
            err = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytesCached{rd.seq, boff, qual}, have_qual, ns);
        } else {
            err = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytesCached{rd.seq, boff, qual}, have_qual, sink);
        }
    }
    if (out.new_pos) out.new_pos[i] = st.pos;
    if (out.new_ncig) out.new_ncig[i] = st.err ? 0u : (uint32_t)st.n;
    if (out.ref_len) out.ref_len[i] = st.err ? 0 : reference_length(cur, st.n);
    if (out.trim_flags) out.trim_flags[i] = st.err ? (uint8_t)0 : (uint8_t)st.flags;
    if (out.status) out.status[i] = (uint8_t)err;
    if (err) atomicAdd(&eb.ctr[2], 1ull);
}

// CIGAR ping-pong in global memory (any length)
__device__ void process_read_serial(const KParams &P, const amp_dev_reads &rd, int64_t i, uint64_t read_base, const DevOut &out,
                                    uint32_t *scratch, uint32_t *counts, const EventBuf &eb, bool status_only) {
    const uint32_t c0 = rd.cig_off32[i];
    const int n = (int)(rd.cig_off32[i + 1] - c0);
    const size_t slot = (size_t)c0 + 3 * (size_t)i;
    DevSink sink{counts, eb, (uint32_t)(read_base + (uint64_t)i)};
    process_read_body(P, rd, i, out, sink, eb, status_only, CigBuf<1>{out.new_cig + slot}, CigBuf<1>{scratch + slot},
                      out.new_cig + slot, c0, n);
}

// Second-pass treatment of a read the tile kernel could not hold (more CIGAR ops than its LDS
// column): serial trim with both CIGAR buffers in LDS columns -- or, for CIGARs of more than D_MAXOPS - 3 ops
// (Nanopore-like reads), ping-pong between the read's output slot and the scratch slot in global memory --, then
//   * regular trimmed CIGAR (clips at the ends, body of M/=/X/I/D/N): deletions / insertion events by
//     the skip-ahead walk on this lane; the match bases are left to the block's waves (count_match_coop)
//     - returns true and leaves the final CIGAR in `cur`;
//   * anything else: the exact serial walk.
constexpr int D_MAXOPS = 19;      // (with the rest of HeavyLds this lets two blocks share a CU's 160 KB)
static_assert(4 * (3 * WV_MAXOPS + WV_STASH_WORDS) <= 2 * D_MAXOPS * 256 && 4 * WV_EVCAP <= 512 && WV_QSTASH <= 528, "the wave path's rows and event stages alias the columns / the block's stage");
template <class CB, class Sink>
__device__ bool process_read_full(const KParams &P, const amp_dev_reads &rd, int64_t i, const DevOut &out, Sink &sink,
                                  const EventBuf &eb, bool status_only, CB &cur, CB &tmp, uint32_t c0, int n,
                                  int &n_final, int32_t &pos_final) {
    uint32_t *const home = out.new_cig + (size_t)c0 + 3 * (size_t)i;
    for (int k = 0; k < n; ++k) cur.set(k, rd.cig[c0 + k]);
    const int32_t lseq = (int32_t)rd.lseq[i];
    const int64_t boff = (int64_t)rd.seq_off8[i] * 8;
    const uint8_t *qual = rd.qual + boff;
    const bool have_qual = lseq > 0 && qual[0] != 0xFF;
    TrimState st{rd.pos[i], n, 0u, 0};
    if (P.do_trim) trim_read_serial(P, st, rd.flag[i], rd.tlen[i], lseq, qual, have_qual, cur, tmp);
    if (!st.err && !is_home(cur, home))
        for (int k = 0; k < st.n; ++k) home[k] = cur.get(k);
    int err = st.err;
    bool coop = false;
    if (!err && P.do_count) {
        NullSink ns;
        if (status_only) {
            err = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytesCached{rd.seq, boff, qual}, have_qual, ns);
        } else {
            bool regular, plain;
            int nseg, e1 = 0, e2 = 0;
            int32_t qs = 0, qe = 0;
            classify(cur, st.n, lseq, regular, plain, nseg);
            if (!have_qual || lseq <= 0) regular = false;
            if (regular) { qs = query_alignment_start(cur, st.n, lseq, e1); qe = query_alignment_end(cur, st.n, lseq, e2); }
            if (regular && !e1 && !e2) {
                if (!plain && count_regular_skip(P, cur, st.n, st.pos, lseq, qs, qe, QualAt{qual}, sink))
                    err = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytesCached{rd.seq, boff, qual}, have_qual, ns);
                coop = err == 0;
            } else {
                err = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytesCached{rd.seq, boff, qual}, have_qual, sink);
            }
        }
    }
    if (out.new_pos) out.new_pos[i] = st.pos;
    if (out.new_ncig) out.new_ncig[i] = st.err ? 0u : (uint32_t)st.n;
    if (out.ref_len) out.ref_len[i] = st.err ? 0 : reference_length(cur, st.n);
    if (out.trim_flags) out.trim_flags[i] = st.err ? (uint8_t)0 : (uint8_t)st.flags;
    if (out.status) out.status[i] = (uint8_t)err;
    if (err) atomicAdd(&eb.ctr[2], 1ull);
    n_final = st.n; pos_final = st.pos;
    return coop;
}

// Match bases of one regular read by a group of D_GROUP lanes: lane = base of the current match op
// (A:718, A:751-753); every op costs one memory round trip, so several reads per wave hide it.
// A base that cannot be counted (code outside ACGTN, position past the table) makes the group's
// first lane run the exact walk for the read's status, like the tile kernel's status-only deferral.
constexpr int D_GROUP = 16;
template <class CB, class Sink>
__device__ void count_match_coop(const KParams &P, const amp_dev_reads &rd, int64_t i, const DevOut &out, Sink &sink,
                                 const EventBuf &eb, const CB &cig, int n, int32_t pos, int lane) {
    const int32_t lseq = (int32_t)rd.lseq[i];
    const int64_t boff = (int64_t)rd.seq_off8[i] * 8;
    const uint8_t *qual = rd.qual + boff;
    const uint32_t G = (uint32_t)P.ref_len;
    int32_t q = 0, r = pos;
    bool bad = false;
    // the ops come D_GROUP at a time, one per lane, and are handed round with shuffles: a CIGAR that lives in global
    // memory (tens of ops) costs a round trip per sixteen ops instead of one per op
    // (a CIGAR in an LDS column is simply read op by op)
    constexpr bool in_lds = std::is_same<CB, LdsCig256>::value;
    const int lane0 = (int)(threadIdx.x & 63u) & ~(D_GROUP - 1);
    for (int k0 = 0; k0 < n; k0 += D_GROUP) {
        uint32_t mine = 0u;
        if (!in_lds) mine = k0 + lane < n ? cig.get(k0 + lane) : 0u;
        const int kn = n - k0 < D_GROUP ? n - k0 : D_GROUP;
        for (int k = 0; k < kn; ++k) {
            const uint32_t v = in_lds ? cig.get(k0 + k) : (uint32_t)__shfl((int)mine, lane0 + k), op = v & 15u;
            const int32_t len = (int32_t)(v >> 4);
            if (is_match_op(op)) {
                for (int32_t j = lane; j < len; j += D_GROUP) {
                    if ((int32_t)qual[q + j] < P.min_quality) continue;
                    const uint32_t col = code_to_col(base_code(rd.seq, boff, q + j));
                    if (col == 0xFFu || (uint32_t)(r + j) >= G) bad = true;
                    else sink.add(r + j, col);
                }
                q += len; r += len;
            } else if (op == OP_I || op == OP_S) q += len;
            else if (op == OP_D || op == OP_N) r += len;
        }
    }
    const uint64_t gmask = ((1ull << D_GROUP) - 1ull) << lane0;
    if ((__ballot(bad) & gmask) && lane == 0) {
        NullSink ns;
        const int err = count_read_walk(P, cig, n, pos, lseq, ReadBytesCached{rd.seq, boff, qual}, true, ns);
        if (out.status) out.status[i] = (uint8_t)err;
        if (err) atomicAdd(&eb.ctr[2], 1ull);
    }
}

// Variant 1: every read on its own lane.  Kept as the simple kernel the tile kernel is
// A/B-checked against on the GPU.
__global__ void __launch_bounds__(256)
k_reads_lane(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *scratch, uint32_t *counts, EventBuf eb) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rd.n_reads) return;
    process_read_serial(P, rd, i, read_base, out, scratch, counts, eb, false);
}

// Deletions / reference skips and insertion events of a read whose match bases the tile kernel
// already counted: the skip-ahead walk over the FINAL CIGAR the tile kernel wrote out (staged in an
// LDS column first: the walk re-reads ops many times).
template <class Sink>
__device__ int process_read_indels(const KParams &P, const amp_dev_reads &rd, int64_t i, const DevOut &out, Sink &sink, lds_u32 *col) {
    const size_t slot = (size_t)rd.cig_off32[i] + 3 * (size_t)i;
    const int n = (int)out.new_ncig[i];
    const int32_t lseq = (int32_t)rd.lseq[i];
    const int32_t pos = out.new_pos[i];
    const uint8_t *qual = rd.qual + (int64_t)rd.seq_off8[i] * 8;
    const QualAt qf{qual};
    int e1 = 0, e2 = 0;
    if (n <= T_MAXOPS) {
        const LdsCig256 cig{col};
        for (int k = 0; k < n; ++k) cig.set(k, out.new_cig[slot + k]);
        const int32_t qs = query_alignment_start(cig, n, lseq, e1), qe = query_alignment_end(cig, n, lseq, e2);
        return count_regular_skip(P, cig, n, pos, lseq, qs, qe, qf, sink);
    }
    const CigBuf<1> cig{out.new_cig + slot};
    const int32_t qs = query_alignment_start(cig, n, lseq, e1), qe = query_alignment_end(cig, n, lseq, e2);
    return count_regular_skip(P, cig, n, pos, lseq, qs, qe, qf, sink);
}


// Heavy half, off the back of the segment: reads the tile kernel could not take at all (more CIGAR ops
// than its LDS column, unusual CIGAR, no room in the tile's segment table) and reads that need
// their exact status.  Lane = read for trimming and indels, lane groups for match bases.
// The work item of a block is a UNIT of D_UNIT consecutive list segments (neighbours on the reference, so
// they can share one LDS window).  SPARSE = true takes the units with at most one round of entries in all and
// runs them as ONE round (a few heavy reads per segment are the common case, and a round per
// segment would be all latency); SPARSE = false takes the other units segment by segment, so that
// a batch made of heavy reads keeps one block per segment.
constexpr int D_UNIT = 8;
struct HeavyLds {
    uint32_t cig[2 * D_MAXOPS * 256];
    uint32_t win[D_PLANES * D_WIN];
    uint32_t coop[3 * 256];
    uint32_t ev[4 * D_EVCAP];
    uint32_t ucnt[D_UNIT + 1];
    uint32_t lng[256];              // reads of this round for the wave path (amp_wave.hpp)
    uint32_t nlong, nslow, wnev[4];
    uint32_t ncoop, nev;
    unsigned long long evbase, mask;
};

template <bool SPARSE>
__device__ __forceinline__ void heavy_pass(HeavyLds &L, const KParams &P, const amp_dev_reads &rd, uint64_t read_base, const DevOut &out,
                                           uint32_t *scratch, uint32_t *counts, const EventBuf &eb, const uint32_t *dlist,
                                           const uint32_t *dcnt, long long tiles_per_block, long long n_seg, long long dcnt_stride,
                                           const uint32_t *segfirst) {
    uint32_t *const s_cig = L.cig, *const s_win = L.win, *const s_coop = L.coop, *const s_ev = L.ev, *const s_ucnt = L.ucnt;
    uint32_t &s_ncoop = L.ncoop, &s_nev = L.nev;
    unsigned long long &s_evbase = L.evbase, &s_mask = L.mask;
    // Few reads are heavy: a small persistent grid looks at the counts of 64 work items per wave-load and only
    // enters those that have entries (a block per segment would cost more in empty launches).
    // work items: whole units (SPARSE) or single segments of the other units
    const int64_t n_item = SPARSE ? (n_seg + D_UNIT - 1) / D_UNIT : n_seg;
    for (int64_t u0 = blockIdx.x; u0 < n_item; u0 += (int64_t)gridDim.x * 64) {
    if (threadIdx.x < 64) {
        const int64_t it = u0 + (int64_t)threadIdx.x * gridDim.x;
        uint32_t tot = 0, own = 0;
        if (it < n_item) {
            const int64_t unit = SPARSE ? it : it / D_UNIT;
            for (int j = 0; j < D_UNIT; ++j) { const int64_t sbj = unit * D_UNIT + j; if (sbj < n_seg) tot += dcnt[5 * dcnt_stride + 64 + sbj]; }
            own = SPARSE ? tot : dcnt[5 * dcnt_stride + 64 + it];
        }
        const unsigned long long mk = __ballot(own != 0 && (tot <= 256u) == SPARSE);
        if (threadIdx.x == 0) s_mask = mk;
    }
    __syncthreads();
    unsigned long long todo = s_mask;
    __syncthreads();
    while (todo) {
    const int64_t item = u0 + (int64_t)(__ffsll((long long)todo) - 1) * gridDim.x;
    todo &= todo - 1;
    {
    const int64_t sb_first = SPARSE ? item * D_UNIT : item;
    __syncthreads();
    if (threadIdx.x == 0) {                      // exclusive prefix of the pass's segment counts
        uint32_t acc = 0;
        for (int j = 0; j < D_UNIT; ++j) {
            s_ucnt[j] = acc;
            const int64_t sbj = sb_first + j;
            if (sbj < n_seg && (SPARSE || j == 0)) acc += dcnt[5 * dcnt_stride + 64 + sbj];
        }
        s_ucnt[D_UNIT] = acc;
        s_ncoop = 0; s_nev = 0;
        L.nlong = 0; L.nslow = 0; L.wnev[0] = L.wnev[1] = L.wnev[2] = L.wnev[3] = 0;
    }
    for (uint32_t k = threadIdx.x; k < D_PLANES * D_WIN; k += blockDim.x) s_win[k] = 0;
    // sorted input: no read of this pass starts left of the first read of its first tile range
    const int64_t first_row = sb_first * tiles_per_block * TILE;
    int32_t base = rd.pos[segfirst ? (int64_t)segfirst[sb_first] : first_row];
    if (base < 0) base = 0;
    __syncthreads();
    const uint32_t cnt = s_ucnt[D_UNIT];
    uint32_t done = 0;   // cooperative entries of earlier rounds (the LDS cursor keeps counting)
    for (uint32_t k0 = 0; k0 < cnt; k0 += blockDim.x) {
        const uint32_t k = k0 + threadIdx.x;
        if (k < cnt) {
            int j = 0;
            if (SPARSE) {
#pragma unroll
                for (int jj = 1; jj < D_UNIT; ++jj) j += k >= s_ucnt[jj] ? 1 : 0;
            }
            const uint32_t *seg_end = dlist + ((size_t)(sb_first + j) + 1) * (size_t)tiles_per_block * TILE;
            const uint32_t e = seg_end[-1 - (int64_t)(k - s_ucnt[j])];
            const int64_t i = (int64_t)(e & DEFER_INDEX_MASK);
            bool status_only = (e & DEFER_STATUS_ONLY) != 0, more = true;
            WinSink sink{(lds_u32 *)s_win, base, counts, eb, (uint32_t)(read_base + (uint64_t)i), (lds_u32 *)s_ev, (lds_u32 *)&s_nev};
            if (e & DEFER_INDELS) {
                if (process_read_indels(P, rd, i, out, sink, (lds_u32 *)s_cig + threadIdx.x)) status_only = true;   // exact status below
                else if (!status_only) more = false;
            }
            if (more) {
                const uint32_t c0 = rd.cig_off32[i];
                const int n = (int)(rd.cig_off32[i + 1] - c0);
                if (n + 3 <= D_MAXOPS) {
                    LdsCig256 cur{(lds_u32 *)s_cig + threadIdx.x}, tmp{(lds_u32 *)s_cig + D_MAXOPS * 256 + threadIdx.x};
                    int nf; int32_t pf;
                    if (process_read_full(P, rd, i, out, sink, eb, status_only, cur, tmp, c0, n, nf, pf)) {
                        // cooperative entry: read, final ops | column of the lane that trimmed it << 22 | buffer B << 30, start
                        const uint32_t slot = atomicAdd(&s_ncoop, 1u) - done;
                        const uint32_t in_b = cur.p != (lds_u32 *)s_cig + threadIdx.x;
                        s_coop[slot * 3] = (uint32_t)i;
                        s_coop[slot * 3 + 1] = (uint32_t)nf | (threadIdx.x << 22) | (in_b << 30);
                        s_coop[slot * 3 + 2] = (uint32_t)pf;
                    }
                } else if (!status_only && n <= WV_MAXOPS - 4) {
                    // tens to hundreds of ops: a wave per read, below
                    L.lng[atomicAdd(&L.nlong, 1u)] = (uint32_t)i;
                } else {
                    // longer CIGARs ping-pong in global memory; counts go through the block's window and the
                    // events into a slice of the list reserved once for the read: its bound is the number of
                    // bases on (q, None) pairs, i.e. of I / P / inner-S ops of the INPUT CIGAR (trimming only
                    // turns such bases into clips)
                    const size_t slot = (size_t)c0 + 3 * (size_t)i;
                    uint32_t bound = 0;
                    if (!status_only && P.do_count) {
                        uint32_t lead = 0, all = 0, trail = 0;
                        bool in_lead = true;
                        for (int k = 0; k < n; ++k) {
                            const uint32_t v = rd.cig[c0 + k], op = v & 15u, len = v >> 4;
                            if (op == OP_H) continue;
                            if (op == OP_S) { all += len; trail += len; if (in_lead) lead += len; }
                            else { in_lead = false; trail = 0; if (op == OP_I || op == OP_P) all += len; }
                        }
                        bound = all - lead - (in_lead ? 0u : trail);
                    }
                    const unsigned shard = blockIdx.x & (EV_SHARDS - 1);
                    unsigned long long base0 = 0;
                    if (bound) base0 = atomicAdd(&eb.ctr[16 + shard], (unsigned long long)bound);
                    const bool fits = bound && (long long)(base0 + bound) <= eb.cap;
                    SliceSink ss{sink, eb.ev + (size_t)shard * (size_t)eb.cap + base0, fits ? bound : 0u, 0u};
                    // the trim runs on this lane; a regular result leaves only its indels here (skip-ahead walk) and
                    // hands the match bases to a group of lanes, which reads the final CIGAR from the output slot
                    CigBuf<1> cur{out.new_cig + slot}, tmp{scratch + slot};
                    int nf; int32_t pf;
                    if (process_read_full(P, rd, i, out, ss, eb, status_only, cur, tmp, c0, n, nf, pf)) {
                        const uint32_t cslot = atomicAdd(&s_ncoop, 1u) - done;
                        s_coop[cslot * 3] = (uint32_t)i;
                        s_coop[cslot * 3 + 1] = ((uint32_t)nf & 0x3FFFFFu) | (1u << 31);      // bit 31: the CIGAR is in global memory
                        s_coop[cslot * 3 + 2] = (uint32_t)pf;
                    }
                    for (uint32_t k = ss.used; k < ss.cap; ++k) ss.slice[k] = amp_ins_event{-1, 0u, 0, 0};
                }
            }
        }
        __syncthreads();
        const uint32_t ncoop = s_ncoop - done;
        for (uint32_t c = threadIdx.x / D_GROUP; c < ncoop; c += blockDim.x / D_GROUP) {
            const int64_t i = (int64_t)s_coop[c * 3];
            const uint32_t w = s_coop[c * 3 + 1];
            WinSink sink{(lds_u32 *)s_win, base, counts, eb, (uint32_t)(read_base + (uint64_t)i), (lds_u32 *)s_ev, (lds_u32 *)&s_nev};
            if (w >> 31) {
                const CigBuf<1> cig{out.new_cig + (size_t)rd.cig_off32[i] + 3 * (size_t)i};
                count_match_coop(P, rd, i, out, sink, eb, cig, (int)(w & 0x3FFFFFu), (int32_t)s_coop[c * 3 + 2], (int)(threadIdx.x % D_GROUP));
            } else {
                const LdsCig256 cig{(lds_u32 *)s_cig + ((w >> 30) & 1u) * (D_MAXOPS * 256) + ((w >> 22) & 0xFFu)};
                count_match_coop(P, rd, i, out, sink, eb, cig, (int)(w & 0x3FFFFFu), (int32_t)s_coop[c * 3 + 2], (int)(threadIdx.x % D_GROUP));
            }
        }
        done += ncoop;
        flush_staged_events(eb, s_ev, &s_nev, &s_evbase);
        if (L.nlong) {
            // wave = read (amp_wave.hpp): three CIGAR rows per wave in the columns' space, events staged per wave
            const uint32_t nlong = L.nlong;
            const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
            lds_u32 *const row = (lds_u32 *)s_cig + wave * (3 * WV_MAXOPS);
            lds_u32 *const wev = (lds_u32 *)s_ev + wave * (WV_EVCAP * 4), *const wn = (lds_u32 *)&L.wnev[wave];
            lds_u8 *const wq = (lds_u8 *)((lds_u32 *)s_cig + 4 * 3 * WV_MAXOPS + wave * WV_STASH_WORDS);
            unsigned long long dummy_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)dummy_acc;
            for (uint32_t c = (uint32_t)wave; c < nlong; c += 4u) {
                const int64_t i = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.lng[c]);
                WaveSink ws{(lds_u32 *)s_win, base, D_WIN, D_WIN, 31u, (uint32_t)AMP_NSYM, counts, eb, (uint32_t)(read_base + (uint64_t)i), wev, wn, (uint32_t)WV_EVCAP};
                if (!wave_read(P, rd, i, wv_header_load(rd, i, lane), out, ws, eb, row, row + WV_MAXOPS, row + 2 * WV_MAXOPS, wq, WV_MAXOPS - 4, lane, dummy_acc)) {
                    if (lane == 0) s_coop[atomicAdd(&L.nslow, 1u)] = (uint32_t)i;      // not a read for the closed forms
                }
                if (*wn > (uint32_t)WV_EVCAP / 2u) wv_flush_events(eb, wev, wn, (uint32_t)WV_EVCAP, lane);
            }
            wv_flush_events(eb, wev, wn, (uint32_t)WV_EVCAP, lane);
            __syncthreads();
            const uint32_t nslow = L.nslow;
            for (uint32_t k = threadIdx.x; k < nslow; k += blockDim.x) {
                // the exact serial code, CIGAR ping-pong between the output slot and the scratch slot
                const int64_t i = (int64_t)s_coop[k];
                const uint32_t c0 = rd.cig_off32[i];
                const size_t slot = (size_t)c0 + 3 * (size_t)i;
                WinSink sink{(lds_u32 *)s_win, base, counts, eb, (uint32_t)(read_base + (uint64_t)i), (lds_u32 *)s_ev, (lds_u32 *)&s_nev};
                process_read_body(P, rd, i, out, sink, eb, false, CigBuf<1>{out.new_cig + slot}, CigBuf<1>{scratch + slot},
                                  out.new_cig + slot, c0, (int)(rd.cig_off32[i + 1] - c0));
            }
            flush_staged_events(eb, s_ev, &s_nev, &s_evbase);
            if (threadIdx.x == 0) { L.nlong = 0; L.nslow = 0; }
            __syncthreads();
        }
    }
    for (uint32_t k = threadIdx.x; k < D_PLANES * D_WIN; k += blockDim.x) {
        const uint32_t v = s_win[k];
        if (!v) continue;
        const uint32_t plane = k / D_WIN, p = (uint32_t)base + k % D_WIN;
        if (plane < AMP_NSYM) atomicAdd(&counts[(size_t)p * AMP_NSYM + plane], v);
        else atomicAdd(&eb.ins_at[p], v);
    }
    __syncthreads();
    }
    }   // items of this round
    }   // rounds
}

// one launch for both kinds of work item: an empty launch is not free when the other step's scan
// kernel holds the LDS of every CU
__global__ void __launch_bounds__(256)
k_deferred_heavy(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *scratch, uint32_t *counts,
                 EventBuf eb, const uint32_t *dlist, const uint32_t *dcnt, long long tiles_per_block, long long n_seg,
                 const GenGeo *geo, long long dcnt_stride, const uint32_t *segfirst) {
    __shared__ HeavyLds L;
    if (__hip_atomic_load(&eb.ctr[24], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) return;   // set by the tile kernel
    if (geo) { tiles_per_block = (long long)geo->tpb; n_seg = (long long)geo->n_seg; }
    heavy_pass<true>(L, P, rd, read_base, out, scratch, counts, eb, dlist, dcnt, tiles_per_block, n_seg, dcnt_stride, segfirst);
    __syncthreads();
    heavy_pass<false>(L, P, rd, read_base, out, scratch, counts, eb, dlist, dcnt, tiles_per_block, n_seg, dcnt_stride, segfirst);
}

// amp_reset: zeroes the device table and the counters
__global__ void __launch_bounds__(256)
k_reset(uint32_t *a, size_t na, uint32_t *b, size_t nb) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = i0 + k;
        if (i < na) a[i] = 0u;
        else if (i - na < nb) b[i - na] = 0u;
    }
}

__global__ void k_add_u32(uint32_t *dst, const uint32_t *src, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// Calling (A:756-771 alleles_from_counts + A:917-952), one lane per reference position.
// Everything that does not depend on the TEXT of an insertion allele is decided here: total
// depth, the six base symbols ranked like sorted(..., reverse=True) (count descending, ties by
// symbol descending: 'T' > 'N' > 'G' > 'C' > 'A' > '-'), consensus symbol, variant record.
// An insertion string can only matter when the position's insertion events could out-rank
// the best base symbol or reach the variant frequency threshold; such positions are flagged
// AMP_CALL_INS_RELEVANT and finished by the host from the event list.
__global__ void __launch_bounds__(256)
k_call(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ ins_at,
                       const uint8_t *__restrict__ ref, int32_t ref_len, amp_call_params pr, amp_pos_call *__restrict__ out,
                       unsigned long long *n_relevant, uint2 *__restrict__ blk) {
    __shared__ uint32_t sv[4], sr[4];
    int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < ref_len;
    if (!live) p = ref_len - 1;            // keep the whole block alive for the block-level counts
    const int desc[6] = {3, 4, 2, 1, 0, 5};            // T N G C A -
    const char sym_of[6] = {'A', 'C', 'G', 'T', 'N', '-'};
    uint32_t c[6], idx[6];
    uint64_t total = ins_at[p];
    for (int k = 0; k < 6; ++k) { idx[k] = desc[k]; c[k] = counts[(size_t)p * AMP_NSYM + desc[k]]; total += c[k]; }
    for (int a = 1; a < 6; ++a) {                      // stable: ties keep the symbol order
        uint32_t cv = c[a], iv = idx[a];
        int b = a - 1;
        while (b >= 0 && c[b] < cv) { c[b + 1] = c[b]; idx[b + 1] = idx[b]; --b; }
        c[b + 1] = cv; idx[b + 1] = iv;
    }
    uint32_t order = 0, nnz = 0;
    for (int k = 0; k < 6; ++k) { order |= idx[k] << (3 * k); nnz += c[k] != 0; }
    amp_pos_call o;
    o.total_depth = (uint32_t)total;
    o.order = order | (nnz << 18);
    o.consensus_sym = -1;
    o.flags = 0; o.alt_mask = 0; o.ref_count = 0;
    const uint32_t ins = ins_at[p];
    const double dtot = (double)total;
    bool relevant = false;
    if (ins) {
        relevant = pr.full_ranking != 0 || ins >= c[0];
        if (pr.run_variants && (double)ins / dtot >= pr.min_freq_variants) relevant = true;
    }
    if (pr.run_consensus && nnz && (int64_t)c[0] >= (int64_t)pr.min_depth_consensus &&
        (double)c[0] / dtot >= pr.min_freq_consensus) o.consensus_sym = (int8_t)idx[0];       // A:928-929
    if (pr.run_variants) {                                                                     // A:933-951
        const char rs = (char)ref[p];
        uint32_t rc = 0; double rf = 0.0; uint32_t n_alt = 0, altm = 0;
        for (int k = 0; k < 6; ++k) {
            if (!c[k]) continue;
            double f = (double)c[k] / dtot;
            if (sym_of[idx[k]] == rs) { rc = c[k]; rf = f; }
            else if (f >= pr.min_freq_variants) { altm |= 1u << k; ++n_alt; }
        }
        o.ref_count = rc; o.alt_mask = (uint8_t)altm;
        if ((int64_t)total >= (int64_t)pr.min_depth_variants && n_alt) o.flags |= AMP_CALL_VARIANT;
        if ((int64_t)rc >= (int64_t)pr.min_depth_variants && rf >= pr.min_freq_variants) o.flags |= AMP_CALL_GT_HAS_REF;
    }
    if (relevant) { o.flags |= AMP_CALL_INS_RELEVANT; if (n_relevant && live) atomicAdd(n_relevant, 1ull); }
    if (live) out[p] = o;
    if (blk) {   // records per 256-position block, for k_call_compact
        const bool isr = live && relevant, isv = live && !relevant && (o.flags & AMP_CALL_VARIANT);
        const unsigned long long bv = __ballot(isv), br = __ballot(isr);
        if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = (uint32_t)__popcll(bv); sr[threadIdx.x >> 6] = (uint32_t)__popcll(br); }
        __syncthreads();
        if (threadIdx.x == 0) blk[blockIdx.x] = make_uint2(sv[0] + sv[1] + sv[2] + sv[3], sr[0] + sr[1] + sr[2] + sr[3]);
    }
}

// Packs the outcome of k_call for the host: consensus column per position, the variant records in
// ascending position (insertion-relevant positions excluded: the host finishes those), and the list
// of insertion-relevant positions.  k_call leaves per-256-position-block record counts; each block
// here places its records behind the totals of the blocks before it.
__global__ void __launch_bounds__(256)
k_call_compact(const amp_pos_call *__restrict__ pc, const uint32_t *__restrict__ counts, int32_t ref_len, const uint2 *__restrict__ blk,
               int8_t *__restrict__ cons, amp_var_rec *__restrict__ vars, int32_t *__restrict__ rel, unsigned long long *n_out) {
    __shared__ uint32_t s_pv[4], s_pr[4], s_wv[4], s_wr[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // totals of the blocks before this one
    uint32_t pv = 0, prl = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) { const uint2 c = blk[b]; pv += c.x; prl += c.y; }
    for (int o = 32; o > 0; o >>= 1) { pv += __shfl_down(pv, o); prl += __shfl_down(prl, o); }
    if (lane == 0) { s_pv[wave] = pv; s_pr[wave] = prl; }
    const int32_t p = blockIdx.x * 256 + tid;
    amp_pos_call c;
    c.flags = 0; c.consensus_sym = -1; c.alt_mask = 0; c.order = 0; c.total_depth = 0; c.ref_count = 0; c.pad = 0;
    if (p < ref_len) { c = pc[p]; cons[p] = c.consensus_sym; }
    const bool isr = (c.flags & AMP_CALL_INS_RELEVANT) != 0, isv = !isr && (c.flags & AMP_CALL_VARIANT);
    const unsigned long long bv = __ballot(isv), br = __ballot(isr);
    if (lane == 0) { s_wv[wave] = (uint32_t)__popcll(bv); s_wr[wave] = (uint32_t)__popcll(br); }
    __syncthreads();
    uint32_t ov = s_pv[0] + s_pv[1] + s_pv[2] + s_pv[3], orl = s_pr[0] + s_pr[1] + s_pr[2] + s_pr[3];
    for (int w = 0; w < wave; ++w) { ov += s_wv[w]; orl += s_wr[w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    ov += (uint32_t)__popcll(bv & below); orl += (uint32_t)__popcll(br & below);
    if (isr) rel[orl] = p;
    if (isv) {
        amp_var_rec v;
        v.pos = p; v.total_depth = c.total_depth; v.ref_count = c.ref_count;
        v.gt_has_ref = (c.flags & AMP_CALL_GT_HAS_REF) ? 1 : 0;
        // ALT slot j takes the j-th ranked symbol that is flagged (static slot indices: a dynamic one puts the record
        // into LDS, 11 KB per block, and the kernel could no longer run next to a block of the fast kernel)
        uint32_t m = c.alt_mask & 0x3Fu;
        uint8_t na = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (m) {
                const int k = __builtin_ctz(m);
                m &= m - 1u;
                const uint32_t col = (c.order >> (3 * k)) & 7u;
                v.alt_col[j] = (uint8_t)col; v.alt_count[j] = counts[(size_t)p * AMP_NSYM + col]; ++na;
            } else {
                v.alt_col[j] = 0xFF; v.alt_count[j] = 0;
            }
        }
        v.n_alt = na;
        vars[ov] = v;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) {   // last thread of the last block knows both totals
        n_out[0] = ov + (isv ? 1u : 0u);
        n_out[1] = orl + (isr ? 1u : 0u);
    }
}

// Copies the text of insertion events (SEQ[q_from:q_to], A:736-738, upper-cased like A:702)
// out of a device-resident batch: one lane per event.
__global__ void k_event_strings(amp_dev_reads rd, uint64_t read_base, int64_t n_ev, const amp_ins_event *__restrict__ ev,
                                const uint64_t *__restrict__ off, uint8_t *__restrict__ text) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ev) return;
    const char nt16[17] = "=ACMGRSVTWYHKDBN";
    const int64_t i = (int64_t)((uint64_t)ev[e].read - read_base);
    const int64_t boff = (int64_t)rd.seq_off8[i] * 8;
    uint8_t *dst = text + off[e];
    for (int32_t q = ev[e].q_from; q < ev[e].q_to; ++q) *dst++ = (uint8_t)nt16[base_code(rd.seq, boff, q)];
}

// ---------------------------------------------------------------------------------------
// library / context API
// ---------------------------------------------------------------------------------------
extern "C" {

int amp_version(void) { return AMP_ABI_VERSION; }

const char *amp_strerror(int rc) {
    switch (rc) {
        case AMP_OK: return "ok";
        case AMP_EINVAL: return "invalid argument";
        case AMP_ENOMEM: return "out of memory";
        case AMP_EHIP: return "HIP runtime error";
        case AMP_ENODEV: return "no usable GPU device";
        case AMP_ESTATE: return "call order violated";
        case AMP_EOVERFLOW: return "output buffer too small";
        case AMP_ERCCL: return "RCCL unavailable or collective failed";
        default: return "unknown error";
    }
}

const char *amp_last_error(const amp_ctx *ctx) { return ctx ? ctx->err : ""; }

const char *amp_read_status_exception(int status) {
    switch (status) {
        case AMP_RS_OK: return "";
        case AMP_RS_INDEX_REF: case AMP_RS_INDEX_PAIRS: case AMP_RS_INDEX_QUERY: case AMP_RS_CIGAR_OP: return "IndexError";
        case AMP_RS_KEY_BASE: return "KeyError";
        case AMP_RS_NO_SEQ: return "AttributeError";
        case AMP_RS_NO_QUAL: case AMP_RS_TYPE: return "TypeError";
        case AMP_RS_CLIP: return "ValueError";
        default: return "RuntimeError";
    }
}

int amp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// A:174-209.  The reference sweeps positions with a deque of the primers whose
// [start-off, end+off) window is open; stale entries never change min/max, so
// the result equals a plain interval cover.  Here: sweep with an explicit active set kept as
// a start-ordered ring, evaluating min/max over it like the reference does.
int amp_find_overlapping_primers(int32_t ref_len, int32_t n, const int32_t *starts, const int32_t *ends, int32_t off,
                                 int32_t *min_start, int32_t *max_end, int32_t *max_primer_len) {
    if (ref_len < 0 || n < 0 || off < 0 || (n && (!starts || !ends)) || (ref_len && (!min_start || !max_end))) return AMP_EINVAL;
    std::vector<int32_t> ring((size_t)std::max(n, 1));
    int head = 0, tail = 0, next = 0;
    for (int32_t p = 0; p < ref_len; ++p) {
        while (head != tail && p >= ends[ring[head]] + off) ++head;
        while (next < n && p >= starts[next] - off) ring[tail++] = next++;
        int32_t mn = -1, mx = -1;
        for (int k = head; k < tail; ++k) {
            int32_t s = starts[ring[k]], e = ends[ring[k]];
            if (k == head || s < mn) mn = s;
            if (k == head || e > mx) mx = e;
        }
        min_start[p] = mn; max_end[p] = mx;
    }
    if (max_primer_len) {
        int32_t m = 0;
        for (int k = 0; k < n; ++k) m = (k == 0) ? ends[k] - starts[k] : std::max(m, ends[k] - starts[k]);
        *max_primer_len = m;
    }
    return AMP_OK;
}

// Layout of the compact calling image (device and pinned host copy alike)
struct CallImage {
    unsigned nblk; size_t off_blk, off_img, img_cons, img_vars, img_rel, img_size;
    explicit CallImage(int32_t G) {
        nblk = (unsigned)((G + 255) / 256);
        off_blk = (size_t)G * sizeof(amp_pos_call);
        off_img = (off_blk + (size_t)nblk * sizeof(uint2) + 63) & ~(size_t)63;
        img_cons = 64; img_vars = img_cons + (((size_t)G + 63) & ~(size_t)63);
        img_rel = img_vars + (size_t)G * sizeof(amp_var_rec); img_size = img_rel + (size_t)G * 4;
    }
};

int amp_ctx_create(amp_ctx **out, int device, int32_t ref_len) {
    if (!out || ref_len <= 0) return AMP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return AMP_ENODEV;
    if (device < 0 || device >= n) return AMP_ENODEV;
    amp_ctx *c = new (std::nothrow) amp_ctx();
    if (!c) return AMP_ENOMEM;
    c->device = device; c->ref_len = ref_len;
    Guard g(c);
    if (!g.ok) { delete c; return AMP_ENODEV; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto fail = [&](int rc) { amp_ctx_destroy(c); return rc; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(AMP_EHIP);
    c->own_stream = true;
    size_t cb = (size_t)ref_len * AMP_DEV_COLS * sizeof(uint32_t);   // counts [G][6] followed by the insertion tally [G]
    if (hipMalloc((void **)&c->d_counts, cb) != hipSuccess) return fail(AMP_ENOMEM);
    c->own_counts = true;
    c->d_ins_at = c->d_counts + (size_t)ref_len * AMP_NSYM;
    if (hipMalloc((void **)&c->d_min_start, (size_t)ref_len * 4) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMalloc((void **)&c->d_max_end, (size_t)ref_len * 4) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMalloc((void **)&c->d_ctr, 32 * sizeof(unsigned long long)) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMalloc((void **)&c->d_ref, (size_t)ref_len) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMemsetAsync(c->d_counts, 0, cb, c->stream) != hipSuccess) return fail(AMP_EHIP);
    if (hipMemsetAsync(c->d_ctr, 0, 32 * sizeof(unsigned long long), c->stream) != hipSuccess) return fail(AMP_EHIP);
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreate(&c->ev2) != hipSuccess || hipEventCreate(&c->ev3) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_call, hipEventDisableTiming) != hipSuccess) return fail(AMP_EHIP);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(AMP_EHIP);
#ifdef AMP_DEV   // development builds only (tools/profile_phases.sh): the shipped library reads no debug switches
    const char *v = getenv("AMPLIHIP_KERNEL");
    if (v && v[0] >= '0' && v[0] <= '5') c->kernel_variant = v[0] - '0';
    v = getenv("AMPLIHIP_PHASES");
    if (v) c->phases = (uint32_t)strtoul(v, nullptr, 0);
#endif
    *out = c;
    return AMP_OK;
}

void amp_ctx_destroy(amp_ctx *c) {
    if (!c) return;
    Guard g(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_counts && c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_min_start) (void)hipFree(c->d_min_start);
    if (c->d_max_end) (void)hipFree(c->d_max_end);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->d_ref) (void)hipFree(c->d_ref);
    DBuf *bufs[] = {&c->events, &c->s_pos, &c->s_flag, &c->s_tlen, &c->s_lseq, &c->s_cigoff, &c->s_cig, &c->s_seqoff,
                    &c->s_seq, &c->s_qual, &c->o_pos, &c->o_ncig, &c->o_cig, &c->o_reflen, &c->o_flags, &c->o_status,
                    &c->scratch, &c->call_buf, &c->agg_buf};
    for (DBuf *b : bufs) b->release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    if (c->ev_call) (void)hipEventDestroy(c->ev_call);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int amp_ctx_set_stream(amp_ctx *c, void *s) {
    if (!c) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return AMP_OK;
}

int amp_ctx_bind_counts(amp_ctx *c, void *dev_counts) {
    if (!c || !dev_counts) return AMP_EINVAL;
    Guard g(c);
    size_t cb = (size_t)c->ref_len * AMP_DEV_COLS * sizeof(uint32_t);
    HIPCHK(c, hipMemcpyAsync(dev_counts, c->d_counts, cb, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_counts) (void)hipFree(c->d_counts);
    c->d_counts = (uint32_t *)dev_counts;
    c->d_ins_at = c->d_counts + (size_t)c->ref_len * AMP_NSYM;
    c->own_counts = false;
    c->call_pending = false;       // (calls begun on the old table are not calls of this one)
    return AMP_OK;
}

int amp_set_primers(amp_ctx *c, const int32_t *mn, const int32_t *mx, int32_t max_primer_len) {
    if (!c || !mn || !mx || max_primer_len < 0) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, hipMemcpyAsync(c->d_min_start, mn, (size_t)c->ref_len * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_max_end, mx, (size_t)c->ref_len * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->max_primer_len = max_primer_len;
    c->have_primers = true;
    return AMP_OK;
}

int amp_set_params(amp_ctx *c, int32_t min_quality, int32_t window, int32_t do_trim, int32_t do_count) {
    if (!c || min_quality < 0 || window < 1) return AMP_EINVAL;  // A:841-844
    c->min_quality = min_quality; c->window = window; c->do_trim = do_trim != 0; c->do_count = do_count != 0;
    return AMP_OK;
}

int amp_set_kernel_variant(amp_ctx *c, int variant) {  // 1 = lane-per-read kernels, 2 = fused tile kernel, 3 = split pipeline, 4 / 5 = fast kernel (first / second generation) + general pass
    if (!c || variant < 0 || variant > 7) return AMP_EINVAL;
    c->kernel_variant = variant;
    return AMP_OK;
}

int amp_fast_path_active(amp_ctx *c) {
    if (!c) return AMP_EINVAL;
    const int kv = c->kernel_variant;
    return ((kv == 0 || kv >= 4) && c->window <= 8 && c->min_quality <= 128) ? 1 : 0;
}

int amp_last_kernel_variant(amp_ctx *c) {
    if (!c) return AMP_EINVAL;
    return c->last_kv;
}

int amp_set_cu_share(amp_ctx *c, int divisor) {
    if (!c || divisor < 1 || divisor > 16) return AMP_EINVAL;
    c->cu_share = divisor;
    return AMP_OK;
}

int amp_set_timing(amp_ctx *c, int split) {
    if (!c) return AMP_EINVAL;
    c->split_timing = split != 0;
    return AMP_OK;
}

int amp_reserve_events(amp_ctx *c, int64_t cap) {
    if (!c || cap < 0) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, grow_events(c, cap));
    c->ev_reserved = true;
    return AMP_OK;
}

int amp_sync(amp_ctx *c) {
    if (!c) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMP_OK;
}

// Launches the read kernels for a device-resident batch on the ctx stream.
static int launch_reads(amp_ctx *c, const amp_dev_reads *rd, uint64_t read_base, const amp_trim_out *o) {
    if (c->do_trim && !c->have_primers) return AMP_ESTATE;
    const int64_t n = rd->n_reads;
    c->timed = false;
    c->call_pending = false;       // (calls begun earlier are for the table as it was)
    if (n == 0) return AMP_OK;
    // event capacity
    if (c->do_count && !c->ev_reserved) {
        unsigned long long h[32];
        HIPCHK(c, hipMemsetAsync(&c->d_ctr[1], 0, sizeof(unsigned long long), c->stream));
        k_event_bound<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(n, rd->cig_off32, rd->cig, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        unsigned long long mx = 0;
        for (int s = 0; s < EV_SHARDS; ++s) mx = std::max(mx, h[16 + s]);
        // any shard may receive every new event; the fast kernel reserves list slots a granule at a time (a refill
        // leaves fewer slots unused than the tile that caused it needs, plus one open granule per wave at the end)
        const FastGrid fgb = fast_grid(n, std::max(1, c->n_cu / c->cu_share));
        // (k_long's waves own granules too: amp_wave.hpp)
        const int64_t need = (int64_t)(mx + 2 * h[1]) + fgb.grid * F_WAVES * (int64_t)F_EVGRAN + 2 * (int64_t)c->n_cu * L_WAVES * L_EVCAP;
        if (need > c->ev_cap) HIPCHK(c, grow_events(c, std::max<int64_t>(need, c->ev_cap + c->ev_cap / 2)));
    }
    KParams P{c->min_quality, c->window, c->do_trim, c->do_count, c->ref_len, c->max_primer_len, c->d_min_start, c->d_max_end, ++c->epoch};
    if (n > 0x7FFFFFFFll) return AMP_EINVAL;
    const size_t slots = (size_t)rd->n_cig + 3 * (size_t)n;
    DevOut out{o ? o->new_pos : nullptr, o ? o->new_ncig : nullptr, o ? o->new_cig : nullptr, o ? o->ref_len : nullptr,
               o ? o->trim_flags : nullptr, o ? o->status : nullptr};
    if (n > (int64_t)DEFER_INDEX_MASK) return AMP_EINVAL;
    // windows wider than a chunk take the serial scan of the general kernel, and the fast kernel's byte-parallel
    // quality test is written for min_quality <= 128: no fast pass for such runs
    // the fast kernel by the batch: its first generation (amp_fast.hpp) keeps a read in registers and is the quicker one for
    // reads of up to 152 bases; the second (amp_fast5.hpp) consumes reads from LDS and takes them up to 304 bases (200 and
    // 250 bp runs: 1.5 x and 1.3 x the first generation, which hands such reads to the general pass)
    const Fast5Cfg f5 = fast5_cfg(n, rd->n_bases_padded, c->window);
    // (a window of 8 makes the first-generation kernel spill 39 registers: 0.354 ms on the bench batch against 0.296 for the second;
    //  windows 5-7 are its own: 0.253 / 0.269 ms at windows of 6 / 7 against 0.277 / 0.280 -- a window of 7 spills 15 registers since the
    //  64-bit adds took four fixed ones, and is still the quicker of the two)
    // (batches of long reads with many CIGAR ops -- three a read and more: soft clips and indels everywhere, BASELINE config 5 --
    //  take the list-driven build of the second generation, amp_fast7.hpp: its tiles hold reads of one length class and none of
    //  the reads that go to the general pass; on batches of uniform long reads it is the slower one, 0.49 against 0.37 ms at 250 bp)
    const bool mixed = f5.waves != 8 && rd->n_cig >= 3 * n;
    const int kv0 = c->kernel_variant == 0 ? (mixed ? 7 : (f5.waves == 8 && c->window != 8) ? 4 : 5) : c->kernel_variant;
    const int kv1 = (kv0 >= 4 && (c->window > 8 || c->min_quality > 128)) ? 2 : kv0;
    const int kv = (kv1 == 6 && c->min_quality < 1) ? 4 : kv1;      // (the third generation tells a masked base by its zeroed code: with min_quality 0 the pad bases of a row would count as kept)
    const int variant = kv >= 5 ? 4 : kv;          // (5 and 6 differ from 4 in the fast kernel only)
    const TileGrid tg = tile_grid(n, c->n_cu);
    const int fast_cus = std::max(1, c->n_cu / c->cu_share);
    const FastGrid fg = kv == 7 ? fast7_grid(n, fast_cus) : kv == 6 ? fast6_grid(n, fast_cus) : kv == 5 ? fast5_grid(n, fast_cus, f5) : fast_grid(n, fast_cus);
    // scratch: [CIGAR ping-pong slots][deferred list][list counts, debug words][variant 3 hand-over][outputs the caller
    // did not ask for but the second pass reads][variant 4: per-block lists, their counts, the dense list, geometry]
    // general pass of variant 4: at most four blocks per CU (its list is usually a tenth of the batch; blocks without
    // tiles would still have to be placed on a CU one after the other), tiles per block decided on the device
    const int64_t gen_grid = std::min<int64_t>(std::min<int64_t>(tg.grid, 4 * (int64_t)c->n_cu), 1024);      // (1024: the words of segfirst, one per block)
    const int64_t n_tiles_max = (n + TILE - 1) / TILE;
    const int64_t gen_tpb_max = (((n_tiles_max + gen_grid - 1) / gen_grid + T_WAVES - 1) / T_WAVES) * T_WAVES;
    const size_t dlist_words = std::max(((size_t)tg.grid + 1) * (size_t)tg.tpb, (size_t)(n_tiles_max + gen_tpb_max + T_WAVES)) * TILE;
    // a batch of reads with many CIGAR ops (eight a read on average: Nanopore-like) gets k_long (amp_wave.hpp) for them; the
    // results do not depend on this choice
    const bool long_kernel = variant == 4 && rd->n_cig >= 8 * n;
    const size_t fast_words = variant == 4 ? (size_t)fg.grid * (size_t)fg.rpb + (size_t)fg.grid * F_WAVES + (size_t)n + 64 + 1024 + (long_kernel ? 2 * (size_t)n : 0) + (kv == 6 ? (size_t)fg.grid * (size_t)fg.rpb : kv == 7 ? 2 * (size_t)fg.grid * (size_t)fg.rpb : 0) : 0;
    HIPCHK(c, c->scratch.ensure((slots * (out.new_cig ? 1 : 2) + (size_t)n * 7 + (size_t)tg.grid * 6 + 64 + dlist_words + fast_words) * 4));
    uint32_t *scr = c->scratch.as<uint32_t>();
    uint32_t *dlist = scr + slots;                                   // one segment of tpb*64 entries per tile-kernel block
    uint32_t *dcnt = dlist + dlist_words;                             // entries used in each segment
    uint32_t *extra = dcnt + tg.grid * 6 + 64;           // [grid] light counts | 64 | [4*grid] debug | [grid] heavy counts
    SplitDesc sd{(int32_t *)extra, extra + n, extra + 2 * n, extra + 3 * n};      // variant 3 hand-over arrays
    extra += 4 * n;
    c->dbg_dcnt = dcnt; c->dbg_grid = (int)tg.grid;
#ifdef AMP_DEV
    if (c->phases & 0x100u) HIPCHK(c, hipMemsetAsync(dcnt, 0, ((size_t)tg.grid * 5 + 64) * 4, c->stream));
#endif
    if (!out.new_pos) { out.new_pos = (int32_t *)extra; }
    extra += n;
    if (!out.new_ncig) { out.new_ncig = extra; }
    extra += n;
    if (!out.new_cig) { out.new_cig = extra; extra += slots; }
    uint32_t *glist = extra, *gcnt = glist + (size_t)fg.grid * (size_t)fg.rpb, *gdense = gcnt + fg.grid * F_WAVES;
    GenGeo *geo = (GenGeo *)(gdense + ((n + 3) & ~(int64_t)3));
    const EventBuf eb{c->events.as<amp_ins_event>(), c->d_ctr, c->d_ins_at, (long long)c->ev_cap};
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (variant == 1) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        k_reads_lane<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(P, *rd, read_base, out, scr, c->d_counts, eb);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev2, c->stream));
    } else if (variant == 4) {
        // fast pass over the simple reads, then the general tile kernel over the list of the others
        const SplitDesc none{nullptr, nullptr, nullptr, nullptr};
        if (c->split_timing) HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        uint32_t *segfirst0 = (uint32_t *)geo + 4;
        uint32_t *clist = segfirst0 + 1024 + (long_kernel ? 2 * (size_t)n : 0);      // (variant 6) the blocks' class lists
        if ((kv == 7 ? fast7_launch(P, *rd, read_base, out, c->d_counts, eb, glist, gcnt, clist, fg, c->stream)
             : kv == 6 ? fast6_launch(P, *rd, read_base, out, c->d_counts, eb, glist, gcnt, clist, fg, c->stream)
             : kv == 5 ? fast5_launch(P, *rd, read_base, out, c->d_counts, eb, glist, gcnt, fg, f5, c->stream)
                     : fast_launch(P, *rd, read_base, out, c->d_counts, eb, glist, gcnt, fg, c->stream, dcnt + tg.grid + 64)) != 0) {
            snprintf(c->err, sizeof(c->err), "fast kernel launch failed"); return AMP_EHIP;
        }
        if (c->split_timing) HIPCHK(c, hipEventRecord(c->ev2, c->stream));
        uint32_t *segfirst = (uint32_t *)geo + 4, *llist = segfirst + 1024, *lpos = llist + n;
        // the list stays in the fast kernel's per-block segments and the tile kernel indexes them itself -- one launch less --
        // unless k_long needs the dense list (to flag its reads in) or the fast kernel ran more blocks than the tile kernel's table holds
        const bool direct = !long_kernel && fg.grid <= GL_MAXSEG;
        const ListSrc ls{direct ? glist : nullptr, direct ? gcnt : nullptr, (int)fg.grid, (int)fg.rpb, (uint32_t)gen_grid, segfirst, geo};
        // (ctr[26..28] -- k_long's list length, its chunk ticket, the entries left to the tile kernel -- are zeroed by the fast kernel)
        if (!direct) {
            k_gcompact<<<(unsigned)fg.grid, 256, 0, c->stream>>>(glist, gcnt, (int)fg.rpb, n, gdense, geo, (uint32_t)gen_grid, c->d_ctr,
                                                                  rd->cig_off32, llist, lpos, long_kernel ? L_MAXOPS - 4 : 0);
            HIPCHK(c, hipGetLastError());
        }
        if (long_kernel) {
            k_long<<<2u * (unsigned)c->n_cu, L_WAVES * 64, 0, c->stream>>>(P, *rd, read_base, out, c->d_counts, eb, llist, lpos, gdense);
            HIPCHK(c, hipGetLastError());
        }
#ifdef AMP_DEV
        if (c->phases & 0x100u) {           // stamps of the general pass alone: the fast kernel's are dropped
            HIPCHK(c, hipMemsetAsync(&c->d_ctr[4], 0, 12 * sizeof(unsigned long long), c->stream));
            k_tile<true, false, true><<<(unsigned)gen_grid, T_WAVES * 64, 0, c->stream>>>(P, *rd, read_base, out, c->d_counts, eb, dlist, dcnt, 0, none,
                                                                                       direct ? nullptr : gdense, geo, (uint32_t)tg.grid, ls AMP_PHASES_ARG(c->phases));
        } else
#endif
        k_tile<false, false, true><<<(unsigned)gen_grid, T_WAVES * 64, 0, c->stream>>>(P, *rd, read_base, out, c->d_counts, eb, dlist, dcnt, 0, none,
                                                                                    direct ? nullptr : gdense, geo, (uint32_t)tg.grid, ls AMP_PHASES_ARG(c->phases));
        HIPCHK(c, hipGetLastError());
        // (the tile kernel does the indels of regular reads itself and raises the heavy pass's flag: k_deferred_light, round 1's
        // second-pass kernel for them, is no longer launched)
        k_deferred_heavy<<<(unsigned)std::min<int64_t>(tg.grid, 2 * (int64_t)c->n_cu), 256, 0, c->stream>>>(
            P, *rd, read_base, out, scr, c->d_counts, eb, dlist, dcnt, 0, 0, geo, (long long)tg.grid, segfirst);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        int rc = variant == 3
                     ? split_launch(P, *rd, read_base, out, c->d_counts, eb, dlist, dcnt, c->n_cu, c->phases, sd, c->stream)
                     : tile_launch(P, *rd, read_base, out, c->d_counts, eb, dlist, dcnt, c->n_cu, c->phases, c->stream);
        if (rc != 0) { snprintf(c->err, sizeof(c->err), "tile kernel launch failed: %s", hipGetErrorString((hipError_t)rc)); return AMP_EHIP; }
        HIPCHK(c, hipEventRecord(c->ev2, c->stream));
        k_deferred_heavy<<<(unsigned)std::min<int64_t>(tg.grid, 2 * (int64_t)c->n_cu), 256, 0, c->stream>>>(
            P, *rd, read_base, out, scr, c->d_counts, eb, dlist, dcnt, (long long)tg.tpb, (long long)tg.grid, nullptr, (long long)tg.grid, nullptr);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev3, c->stream));
    c->timed = true;
    c->last_split = variant != 4 || c->split_timing;
    c->last_kv = kv;
    return AMP_OK;
}

int amp_process_batch_device(amp_ctx *c, const amp_dev_reads *rd, uint64_t read_base, const amp_trim_out *dev_out) {
    if (!c || !rd || rd->n_reads < 0) return AMP_EINVAL;
    if (rd->n_reads && (!rd->pos || !rd->flag || !rd->tlen || !rd->lseq || !rd->cig_off32 || !rd->cig || !rd->seq_off8 ||
                        !rd->seq || !rd->qual)) return AMP_EINVAL;
    Guard g(c);
    c->staged_n = -1;          // (amp_event_strings(reads = NULL) refers to the last HOST batch: there is none now)
    return launch_reads(c, rd, read_base, dev_out);
}

int amp_process_batch(amp_ctx *c, const amp_reads *r, uint64_t read_base, const amp_trim_out *out) {
    if (!c || !r || r->n_reads < 0) return AMP_EINVAL;
    const int64_t n = r->n_reads;
    if (n == 0) return AMP_OK;
    if (!r->pos || !r->flag || !r->tlen || !r->lseq || !r->cig_off || !r->cig || !r->seq_off || !r->seq || !r->qual) return AMP_EINVAL;
    Guard g(c);
    const uint64_t n_cig = r->cig_off[n], n_bases = r->seq_off[n];
    if (n_cig > 0xFFFFFFF0ull || (n_bases >> 3) > 0xFFFFFFF0ull || (n_bases & 7)) return AMP_EINVAL;
    std::vector<uint32_t> co((size_t)n + 1), so((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) {
        if (r->seq_off[i] & 7) return AMP_EINVAL;
        if (i < n && (r->cig_off[i + 1] < r->cig_off[i] || r->seq_off[i + 1] < r->seq_off[i] + r->lseq[i])) return AMP_EINVAL;
        co[(size_t)i] = (uint32_t)r->cig_off[i];
        so[(size_t)i] = (uint32_t)(r->seq_off[i] >> 3);
    }
    hipStream_t s = c->stream;
    struct Up { DBuf *b; const void *src; size_t bytes; };
    Up ups[] = {{&c->s_pos, r->pos, (size_t)n * 4}, {&c->s_flag, r->flag, (size_t)n * 2}, {&c->s_tlen, r->tlen, (size_t)n * 4},
                {&c->s_lseq, r->lseq, (size_t)n * 4}, {&c->s_cigoff, co.data(), ((size_t)n + 1) * 4},
                {&c->s_cig, r->cig, (size_t)n_cig * 4}, {&c->s_seqoff, so.data(), ((size_t)n + 1) * 4},
                {&c->s_seq, r->seq, (size_t)(n_bases / 2)}, {&c->s_qual, r->qual, (size_t)n_bases}};
    for (Up &u : ups) {
        HIPCHK(c, u.b->ensure(u.bytes + 16));  // +16: tile loads may read one vector past the end
        if (u.bytes) HIPCHK(c, hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, s));
    }
    const size_t slots = (size_t)n_cig + 3 * (size_t)n;
    HIPCHK(c, c->o_pos.ensure((size_t)n * 4)); HIPCHK(c, c->o_ncig.ensure((size_t)n * 4));
    HIPCHK(c, c->o_cig.ensure(slots * 4)); HIPCHK(c, c->o_reflen.ensure((size_t)n * 4));
    HIPCHK(c, c->o_flags.ensure((size_t)n)); HIPCHK(c, c->o_status.ensure((size_t)n));
    amp_dev_reads rd{n, c->s_pos.as<int32_t>(), c->s_flag.as<uint16_t>(), c->s_tlen.as<int32_t>(), c->s_lseq.as<uint32_t>(),
                     c->s_cigoff.as<uint32_t>(), c->s_cig.as<uint32_t>(), c->s_seqoff.as<uint32_t>(), c->s_seq.as<uint8_t>(),
                     c->s_qual.as<uint8_t>(), (int64_t)n_cig, (int64_t)n_bases};
    amp_trim_out dout{c->o_pos.as<int32_t>(), c->o_ncig.as<uint32_t>(), c->o_cig.as<uint32_t>(), c->o_reflen.as<int32_t>(),
                      c->o_flags.as<uint8_t>(), c->o_status.as<uint8_t>()};
    c->staged_n = n; c->staged_ncig = (int64_t)n_cig; c->staged_nbases = (int64_t)n_bases;
    int rc = launch_reads(c, &rd, read_base, &dout);
    if (rc != AMP_OK) return rc;
    if (out) {
        if (out->new_pos) HIPCHK(c, hipMemcpyAsync(out->new_pos, dout.new_pos, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        if (out->new_ncig) HIPCHK(c, hipMemcpyAsync(out->new_ncig, dout.new_ncig, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        if (out->new_cig) HIPCHK(c, hipMemcpyAsync(out->new_cig, dout.new_cig, slots * 4, hipMemcpyDeviceToHost, s));
        if (out->ref_len) HIPCHK(c, hipMemcpyAsync(out->ref_len, dout.ref_len, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        if (out->trim_flags) HIPCHK(c, hipMemcpyAsync(out->trim_flags, dout.trim_flags, (size_t)n, hipMemcpyDeviceToHost, s));
        if (out->status) HIPCHK(c, hipMemcpyAsync(out->status, dout.status, (size_t)n, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return AMP_OK;
}

int amp_last_kernel_ms(amp_ctx *c, float *total_ms, float *scan_ms) {
    if (!c) return AMP_EINVAL;
    if (!c->timed) return AMP_ESTATE;
    Guard g(c);
    HIPCHK(c, hipEventSynchronize(c->ev3));
    float t = 0, s = 0;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev0, c->ev3));
    if (!c->last_split) s = t;     // (the first kernel was not timed separately)
    else HIPCHK(c, hipEventElapsedTime(&s, c->ev1, c->ev2));
    if (total_ms) *total_ms = t;
    if (scan_ms) *scan_ms = s;
    return AMP_OK;
}

int amp_get_counts(amp_ctx *c, uint32_t *counts) {
    if (!c || !counts) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, hipMemcpyAsync(counts, c->d_counts, (size_t)c->ref_len * AMP_NSYM * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMP_OK;
}

int amp_add_counts(amp_ctx *c, const uint32_t *counts) {
    if (!c || !counts) return AMP_EINVAL;
    Guard g(c);
    size_t n = (size_t)c->ref_len * AMP_NSYM;
    c->call_pending = false;       // the table changes: calls begun earlier are for the table as it was
    HIPCHK(c, c->call_buf.ensure(n * 4));
    HIPCHK(c, hipMemcpyAsync(c->call_buf.p, counts, n * 4, hipMemcpyHostToDevice, c->stream));
    k_add_u32<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(c->d_counts, c->call_buf.as<uint32_t>(), (int64_t)n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMP_OK;
}

void *amp_counts_device_ptr(amp_ctx *c) { return c ? c->d_counts : nullptr; }

int amp_get_ins_events(amp_ctx *c, int64_t *n, amp_ins_event *buf, int64_t cap) {
    if (!c || !n) return AMP_EINVAL;
    Guard g(c);
    unsigned long long h[EV_SHARDS];
    HIPCHK(c, hipMemcpyAsync(h, &c->d_ctr[16], sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t total = 0;
    bool dropped = false;
    for (int s = 0; s < EV_SHARDS; ++s) { total += (int64_t)h[s]; if ((int64_t)h[s] > c->ev_cap) dropped = true; }
    *n = total;
    if (dropped) return AMP_EOVERFLOW;   // a reserved buffer was too small: events were dropped
    if (buf) {
        if (total > cap) return AMP_EOVERFLOW;
        int64_t o = 0;
        for (int s = 0; s < EV_SHARDS; ++s) {
            if (h[s]) HIPCHK(c, hipMemcpyAsync(buf + o, c->events.as<amp_ins_event>() + (size_t)s * c->ev_cap,
                                               (size_t)h[s] * sizeof(amp_ins_event), hipMemcpyDeviceToHost, c->stream));
            o += (int64_t)h[s];
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int64_t w = 0;                                   // slots a read reserved and did not use carry ref_pos = -1
        for (int64_t k = 0; k < total; ++k)
            if (buf[k].ref_pos >= 0) { if (w != k) buf[w] = buf[k]; ++w; }
        *n = w;
    }
    return AMP_OK;
}

int amp_drain_ins_events(amp_ctx *c, int64_t *n, amp_ins_event *buf, int64_t cap) {
    // amp_get_ins_events, then the list starts over (the per-position tally stays): a run of many batches reads every
    // event once instead of the whole accumulated list after each batch, and the device list stays one batch long
    const int rc = amp_get_ins_events(c, n, buf, cap);
    if (rc != AMP_OK || !buf) return rc;
    Guard g(c);
    HIPCHK(c, hipMemsetAsync(&c->d_ctr[16], 0, EV_SHARDS * sizeof(unsigned long long), c->stream));
    return AMP_OK;
}

int amp_aggregate_ins_events(amp_ctx *c, const amp_dev_reads *rd, uint64_t read_base, int drain, int64_t *n_runs, amp_ins_run *buf, int64_t cap) {
    if (!c || !n_runs) return AMP_EINVAL;
    Guard g(c);
    unsigned long long h[EV_SHARDS];
    HIPCHK(c, hipMemcpyAsync(h, &c->d_ctr[16], sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t total = 0;
    for (int s = 0; s < EV_SHARDS; ++s) { if ((int64_t)h[s] > c->ev_cap) return AMP_EOVERFLOW; total += (int64_t)h[s]; }
    *n_runs = total;
    if (!buf) return AMP_OK;
    if (total > cap) return AMP_EOVERFLOW;
    amp_dev_reads staged;
    if (!rd) {           // the batch of the last amp_process_batch call: its device copy is still in the ctx's staging buffers
        if (c->staged_n < 0) return AMP_ESTATE;
        staged = amp_dev_reads{c->staged_n, c->s_pos.as<int32_t>(), c->s_flag.as<uint16_t>(), c->s_tlen.as<int32_t>(), c->s_lseq.as<uint32_t>(),
                               c->s_cigoff.as<uint32_t>(), c->s_cig.as<uint32_t>(), c->s_seqoff.as<uint32_t>(), c->s_seq.as<uint8_t>(),
                               c->s_qual.as<uint8_t>(), c->staged_ncig, c->staged_nbases};
        rd = &staged;
    }
    int64_t n_ev = 0, nr = 0;
    if (total) {
        const size_t sb = amp::ins_scratch_bytes(total), rb = (((size_t)total * sizeof(amp_ins_run)) + 255) & ~(size_t)255;
        HIPCHK(c, c->agg_buf.ensure(sb + rb));
        amp_ins_run *d_runs = (amp_ins_run *)c->agg_buf.p;
        const int rc = amp::ins_aggregate(c->stream, *rd, read_base, c->events.as<amp_ins_event>(), (long long)c->ev_cap, h,
                                          (uint8_t *)c->agg_buf.p + rb, d_runs, &n_ev, &nr);
        if (rc != 0) { snprintf(c->err, sizeof(c->err), "insertion-event aggregation failed: %d", rc); return AMP_EHIP; }
        if (nr) HIPCHK(c, hipMemcpyAsync(buf, d_runs, (size_t)nr * sizeof(amp_ins_run), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *n_runs = nr;
    if (drain) HIPCHK(c, hipMemsetAsync(&c->d_ctr[16], 0, EV_SHARDS * sizeof(unsigned long long), c->stream));
    return AMP_OK;
}

int amp_debug_blocks(amp_ctx *c, uint32_t *out, int cap_blocks, int *n_blocks) {   // [block][dur, rebases, p2 chunks, p4 chunks]
    if (!c || !out || !n_blocks || !c->dbg_dcnt) return AMP_EINVAL;
    Guard g(c);
    int nb = c->dbg_grid < cap_blocks ? c->dbg_grid : cap_blocks;
    HIPCHK(c, hipMemcpyAsync(out, c->dbg_dcnt + c->dbg_grid + 64, (size_t)nb * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_blocks = nb;
    return AMP_OK;
}

int amp_debug_counters(amp_ctx *c, uint64_t *out16) {  // raw device counters (development aid)
    if (!c || !out16) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, hipMemcpyAsync(out16, c->d_ctr, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->kernel_variant != 1 && c->dbg_dcnt && c->dbg_grid > 0) {      // [3]: deferred reads of the LAST batch, from the per-block list counts
        std::vector<uint32_t> h((size_t)c->dbg_grid * 6 + 64);
        HIPCHK(c, hipMemcpyAsync(h.data(), c->dbg_dcnt, h.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint64_t tot = 0, heavy = 0;
        for (int b = 0; b < c->dbg_grid; ++b) { tot += h[(size_t)b]; heavy += h[(size_t)c->dbg_grid * 5 + 64 + (size_t)b]; }
        out16[3] = tot + heavy;
        out16[0] = heavy;                         // [0]: of which heavy entries (whole read / exact status)
    }
    return AMP_OK;
}

int amp_error_reads(amp_ctx *c, int64_t *n) {  // reads with a non-zero status since the last reset
    if (!c || !n) return AMP_EINVAL;
    Guard g(c);
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, &c->d_ctr[2], sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n = (int64_t)h;
    return AMP_OK;
}

int amp_reset(amp_ctx *c) {
    if (!c) return AMP_EINVAL;
    Guard g(c);
    c->call_pending = false;
    // one small kernel for the table and the counters (two hipMemsetAsync calls are three fill kernels of 5 us each)
    const size_t words = (size_t)c->ref_len * AMP_DEV_COLS;
    k_reset<<<(unsigned)((words + 64 + 1023) / 1024), 256, 0, c->stream>>>(c->d_counts, words, (uint32_t *)c->d_ctr, 64);
    HIPCHK(c, hipGetLastError());
    return AMP_OK;
}

// RCCL is resolved at run time so the library has no link-time dependency on it and uses
// whichever librccl the process (e.g. torch) already loaded.
typedef int (*nccl_reduce_fn)(const void *, void *, size_t, int, int, int, void *, hipStream_t);
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
int amp_reduce(amp_ctx *c, void *comm, int root) {
    if (!c) return AMP_EINVAL;
    if (!comm) return AMP_OK;
    Guard g(c);
    c->call_pending = false;       // the table changes: calls begun earlier are for the un-reduced table
    void *h = dlopen(nullptr, RTLD_NOW);
    void *f_red = h ? dlsym(h, "ncclReduce") : nullptr;
    void *f_all = h ? dlsym(h, "ncclAllReduce") : nullptr;
    if (!f_red || !f_all) {
        void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) { f_red = dlsym(lib, "ncclReduce"); f_all = dlsym(lib, "ncclAllReduce"); }
    }
    if (!f_red || !f_all) { snprintf(c->err, sizeof(c->err), "RCCL symbols not found"); return AMP_ERCCL; }
    const size_t cnt = (size_t)c->ref_len * AMP_DEV_COLS;
    const int nccl_uint32 = 3, nccl_sum = 0;
    int rc = root < 0 ? ((nccl_allreduce_fn)f_all)(c->d_counts, c->d_counts, cnt, nccl_uint32, nccl_sum, comm, c->stream)
                      : ((nccl_reduce_fn)f_red)(c->d_counts, c->d_counts, cnt, nccl_uint32, nccl_sum, root, comm, c->stream);
    if (rc != 0) { snprintf(c->err, sizeof(c->err), "RCCL reduce failed: %d", rc); return AMP_ERCCL; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMP_OK;
}

// ---------------------------------------------------------------------------------------
// calling (A:756-771, A:917-952)
// ---------------------------------------------------------------------------------------
int amp_set_reference(amp_ctx *c, const uint8_t *ref_ascii) {
    if (!c || !ref_ascii) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, hipMemcpyAsync(c->d_ref, ref_ascii, (size_t)c->ref_len, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_ref = true;
    return AMP_OK;
}

int amp_call_positions(amp_ctx *c, const amp_call_params *pr, amp_pos_call *out, int64_t *n_relevant) {
    if (!c || !pr || !out) return AMP_EINVAL;
    if (pr->run_variants && !c->have_ref) return AMP_ESTATE;
    Guard g(c);
    const int32_t G = c->ref_len;
    HIPCHK(c, c->call_buf.ensure((size_t)G * sizeof(amp_pos_call)));
    HIPCHK(c, hipMemsetAsync(&c->d_ctr[4], 0, sizeof(unsigned long long), c->stream));
    k_call<<<(unsigned)((G + 255) / 256), 256, 0, c->stream>>>(c->d_counts, c->d_ins_at, c->d_ref, G, *pr,
                                                              c->call_buf.as<amp_pos_call>(), &c->d_ctr[4], nullptr);
    HIPCHK(c, hipGetLastError());
    unsigned long long nr = 0;
    HIPCHK(c, hipMemcpyAsync(out, c->call_buf.p, (size_t)G * sizeof(amp_pos_call), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&nr, &c->d_ctr[4], sizeof(nr), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_relevant) *n_relevant = (int64_t)nr;
    return AMP_OK;
}

// Enqueues the calling kernels on the ctx stream; does not wait.  (The caller has read the image of the ctx's previous call:
// the views are valid until the next call_* on the ctx.)
static int call_compact_enqueue(amp_ctx *c, const amp_call_params *pr, bool pinned) {
    const int32_t G = c->ref_len;
    // device buffer: [per-position calls][block counts]( + the image when it is copied); host: the output image [totals 64 B]
    // [consensus][records][relevant positions]
    const CallImage L(G);
    HIPCHK(c, c->call_buf.ensure(L.off_img + (pinned ? 64 : L.img_size)));
    uint8_t *base = c->call_buf.as<uint8_t>();
    amp_pos_call *d_pc = (amp_pos_call *)base;
    uint8_t *hz;
    if (pinned) {
        // Pipelined callers (amp_call_compact_begin): the result image is written by the compaction kernel straight into a
        // pinned host buffer (plain stores over the host link; half a megabyte): no copy to place.  A copy on a stream of its
        // own was only served once the kernels queued behind the calls had drained (0.61 ms per step with three steps in flight
        // instead of 0.31), a copy in the work stream cost a launch gap more (0.308 -> 0.294 ms per bench step with this).
        if (c->h_pin_cap < L.img_size) {
            if (c->h_pin) (void)hipHostFree(c->h_pin);
            c->h_pin = nullptr; c->h_pin_cap = 0;
            HIPCHK(c, hipHostMalloc(&c->h_pin, L.img_size, hipHostMallocDefault));
            c->h_pin_cap = L.img_size;
        }
        hz = (uint8_t *)c->h_pin;
    } else {
        // One-shot callers (a command line: one view per run): no page-locked memory.  In a process that has just mapped and
        // released hundreds of MB of host memory (a BAM inflated piece by piece) hipHostMalloc / hipHostFree of this 1.5 MB
        // image took 25-70 ms each (rocprofv3 --hip-trace), a third of a `variants` run over 1.5 M reads; a device image and
        // one 1.5 MB copy take 0.3 ms.
        hz = base + L.off_img;
        if (c->h_img.size() < L.img_size) c->h_img.resize(L.img_size);
    }
    k_call<<<L.nblk, 256, 0, c->stream>>>(c->d_counts, c->d_ins_at, c->d_ref, G, *pr, d_pc, nullptr, (uint2 *)(base + L.off_blk));
    HIPCHK(c, hipGetLastError());
    k_call_compact<<<L.nblk, 256, 0, c->stream>>>(d_pc, c->d_counts, G, (const uint2 *)(base + L.off_blk), (int8_t *)(hz + L.img_cons),
                                                  (amp_var_rec *)(hz + L.img_vars), (int32_t *)(hz + L.img_rel), (unsigned long long *)hz);
    HIPCHK(c, hipGetLastError());
    if (!pinned) HIPCHK(c, hipMemcpyAsync(c->h_img.data(), hz, L.img_size, hipMemcpyDeviceToHost, c->stream));
    c->img_pinned = pinned;
    HIPCHK(c, hipEventRecord(c->ev_call, c->stream));     // "the image has arrived"
    return AMP_OK;
}

int amp_call_compact_begin(amp_ctx *c, const amp_call_params *pr) {
    if (!c || !pr) return AMP_EINVAL;
    if (pr->run_variants && !c->have_ref) return AMP_ESTATE;
    Guard g(c);
    const int rc = call_compact_enqueue(c, pr, true);
    if (rc != AMP_OK) return rc;
    c->call_pending = true; c->call_pending_params = *pr;     // (the view waits for the copy's event, not for what is enqueued behind it)
    return AMP_OK;
}

int amp_call_compact_view(amp_ctx *c, const amp_call_params *pr, amp_call_view *view) {
    if (!c || !pr || !view) return AMP_EINVAL;
    if (pr->run_variants && !c->have_ref) return AMP_ESTATE;
    Guard g(c);
    const int32_t G = c->ref_len;
    const CallImage L(G);
    // work enqueued by amp_call_compact_begin with the same parameters is picked up here; anything else starts now
    const bool begun = c->call_pending && memcmp(&c->call_pending_params, pr, sizeof(*pr)) == 0;
    c->call_pending = false;
    if (!begun) { const int rc = call_compact_enqueue(c, pr, false); if (rc != AMP_OK) return rc; }
    uint8_t *hp = c->img_pinned ? (uint8_t *)c->h_pin : c->h_img.data();
    const unsigned long long *h_nn = (const unsigned long long *)hp;
    const int8_t *h_cons = (const int8_t *)(hp + L.img_cons);
    amp_var_rec *h_vars = (amp_var_rec *)(hp + L.img_vars);
    int32_t *h_rel = (int32_t *)(hp + L.img_rel);
    HIPCHK(c, hipEventSynchronize(c->ev_call));
    const int64_t nv = (int64_t)h_nn[0], nr = (int64_t)h_nn[1];
    *view = amp_call_view{h_cons, h_vars, h_rel, nv, nr};
    return AMP_OK;
}

int amp_call_compact(amp_ctx *c, const amp_call_params *pr, int8_t *consensus, amp_var_rec *vars, int64_t vars_cap,
                     int64_t *n_vars, int32_t *relevant, int64_t relevant_cap, int64_t *n_relevant) {
    if (!c || !pr || !consensus || !n_vars || !n_relevant || vars_cap < 0 || relevant_cap < 0) return AMP_EINVAL;
    amp_call_view v;
    const int rc = amp_call_compact_view(c, pr, &v);
    if (rc != AMP_OK) return rc;
    *n_vars = v.n_vars; *n_relevant = v.n_relevant;
    if (v.n_vars > vars_cap || v.n_relevant > relevant_cap) return AMP_EOVERFLOW;
    memcpy(consensus, v.consensus, (size_t)c->ref_len);
    if (v.n_vars && vars) memcpy(vars, v.vars, (size_t)v.n_vars * sizeof(amp_var_rec));
    if (v.n_relevant && relevant) memcpy(relevant, v.relevant, (size_t)v.n_relevant * 4);
    return AMP_OK;
}

// The reference's coordinate helpers on the device, one lane per case: get_pos_on_query (A:389-412), get_pos_on_ref
// (A:363-386) and fix_cigar (A:415-423) -- the very device functions the kernels use (amp_read.hpp), driven directly.
__global__ void __launch_bounds__(256)
k_coordinate_helpers(int64_t n, const uint32_t *__restrict__ cig_off, uint32_t *cig, const int32_t *__restrict__ ref_start,
                     const int32_t *__restrict__ ref_pos, const int32_t *__restrict__ query_pos, int32_t *out_q, int32_t *out_r,
                     uint32_t *fixed, uint32_t *fixed_n, uint8_t *status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c0 = cig_off[i];
    const int nops = (int)(cig_off[i + 1] - c0);
    const CigBuf<1> c{cig + c0};
    int e1 = 0, e2 = 0;
    out_q[i] = pos_on_query(c, nops, ref_pos[i], ref_start[i], e1);
    out_r[i] = pos_on_ref(c, nops, query_pos[i], ref_start[i], e2);
    Emitter<CigBuf<1>> e{CigBuf<1>{fixed + c0}};
    for (int k = 0; k < nops; ++k) { const uint32_t v = c.get(k); e.push(v & 15u, v >> 4); }
    fixed_n[i] = (uint32_t)e.finish();
    status[i] = (uint8_t)((e1 & 15) | ((e2 & 15) << 4));      // one nibble per helper: the reference's get_pos_on_query returns before it touches later ops
}

int amp_coordinate_helpers(amp_ctx *c, int64_t n, const uint32_t *cig_off, const uint32_t *cig, const int32_t *ref_start,
                           const int32_t *ref_pos, const int32_t *query_pos, int32_t *pos_on_query_out, int32_t *pos_on_ref_out,
                           uint32_t *fixed_cig, uint32_t *fixed_n, uint8_t *status) {
    if (!c || n < 0) return AMP_EINVAL;
    if (n == 0) return AMP_OK;
    if (!cig_off || !ref_start || !ref_pos || !query_pos || !pos_on_query_out || !pos_on_ref_out || !fixed_cig || !fixed_n || !status)
        return AMP_EINVAL;
    const size_t nc = cig_off[n];
    if (nc && !cig) return AMP_EINVAL;
    Guard g(c);
    // one scratch image: offsets | ops | starts | ref positions | query positions | the two results | fixed ops | their counts | status
    const size_t words = ((size_t)n + 1) + nc + 3 * (size_t)n + 2 * (size_t)n + nc + (size_t)n + ((size_t)n + 3) / 4 + 16;
    HIPCHK(c, c->call_buf.ensure(words * 4));
    uint32_t *d = c->call_buf.as<uint32_t>();
    uint32_t *d_off = d, *d_cig = d_off + n + 1;
    int32_t *d_rs = (int32_t *)(d_cig + nc), *d_rp = d_rs + n, *d_qp = d_rp + n, *d_oq = d_qp + n, *d_or = d_oq + n;
    uint32_t *d_fx = (uint32_t *)(d_or + n), *d_fn = d_fx + nc;
    uint8_t *d_st = (uint8_t *)(d_fn + n);
    HIPCHK(c, hipMemcpyAsync(d_off, cig_off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (nc) HIPCHK(c, hipMemcpyAsync(d_cig, cig, nc * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rs, ref_start, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rp, ref_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_qp, query_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    k_coordinate_helpers<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(n, d_off, d_cig, d_rs, d_rp, d_qp, d_oq, d_or, d_fx, d_fn, d_st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(pos_on_query_out, d_oq, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(pos_on_ref_out, d_or, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (nc) HIPCHK(c, hipMemcpyAsync(fixed_cig, d_fx, nc * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fixed_n, d_fn, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, d_st, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMP_OK;
}

int amp_event_strings(amp_ctx *c, const amp_dev_reads *rd, uint64_t read_base, int64_t n_ev, const amp_ins_event *ev,
                      const uint64_t *off, uint8_t *text) {
    if (!c || n_ev < 0 || (n_ev && (!ev || !off || !text))) return AMP_EINVAL;
    if (n_ev == 0) return AMP_OK;
    amp_dev_reads staged;
    if (!rd) {           // the batch of the last amp_process_batch call: its device copy is still in the ctx's staging buffers
        if (c->staged_n < 0) return AMP_ESTATE;
        staged = amp_dev_reads{c->staged_n, c->s_pos.as<int32_t>(), c->s_flag.as<uint16_t>(), c->s_tlen.as<int32_t>(), c->s_lseq.as<uint32_t>(),
                               c->s_cigoff.as<uint32_t>(), c->s_cig.as<uint32_t>(), c->s_seqoff.as<uint32_t>(), c->s_seq.as<uint8_t>(),
                               c->s_qual.as<uint8_t>(), c->staged_ncig, c->staged_nbases};
        rd = &staged;
    }
    for (int64_t e = 0; e < n_ev; ++e) {
        uint64_t i = (uint64_t)ev[e].read - read_base;
        if (i >= (uint64_t)rd->n_reads || ev[e].q_from < 0 || ev[e].q_to < ev[e].q_from ||
            off[e + 1] - off[e] != (uint64_t)(ev[e].q_to - ev[e].q_from)) return AMP_EINVAL;
    }
    Guard g(c);
    const size_t tb = (size_t)off[n_ev];
    const size_t need = (size_t)n_ev * sizeof(amp_ins_event) + ((size_t)n_ev + 1) * 8 + tb + 64;
    HIPCHK(c, c->call_buf.ensure(need));
    uint8_t *base = c->call_buf.as<uint8_t>();
    uint64_t *d_off = (uint64_t *)base;
    amp_ins_event *d_ev = (amp_ins_event *)(base + ((size_t)n_ev + 1) * 8);
    uint8_t *d_text = (uint8_t *)(d_ev + n_ev);
    HIPCHK(c, hipMemcpyAsync(d_off, off, ((size_t)n_ev + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_ev, ev, (size_t)n_ev * sizeof(amp_ins_event), hipMemcpyHostToDevice, c->stream));
    k_event_strings<<<(unsigned)((n_ev + 255) / 256), 256, 0, c->stream>>>(*rd, read_base, n_ev, d_ev, d_off, d_text);
    HIPCHK(c, hipGetLastError());
    if (tb) HIPCHK(c, hipMemcpyAsync(text, d_text, tb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMP_OK;
}

}  // extern "C"
