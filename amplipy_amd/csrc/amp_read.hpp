// amp_read.hpp -- per-read device logic of libamplihip: CIGAR coordinate maps, primer /
// quality soft-clipping, and the exact aligned-pair walk used for reads the tile kernel
// does not take on its fast path.
//
// Every function restates behaviour of /root/reference/AmpliPy.py v0.0.2 (cited A:line);
// none of it is a translation: the reference builds Python lists and calls fix_cigar after
// each stage, here each stage streams ops through an emitter that merges equal neighbours
// on the fly, works on `len<<4|op` words in LDS or global memory, and keeps all state in
// registers.  Functions are __host__ __device__ only so that tests/hostsim can run the same
// code on the CPU against the golden vectors; the shipped library never executes them on
// the host.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amplihip.h"

#define AMP_HD __host__ __device__ __forceinline__

namespace amp {

constexpr uint32_t OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8;

// A:43-44 as bit sets over op codes 0..8
AMP_HD bool consumes_query(uint32_t op) { return (0x193u >> op) & 1u; }  // M I S = X
AMP_HD bool consumes_ref(uint32_t op) { return (0x18Du >> op) & 1u; }    // M D N = X
AMP_HD bool is_match_op(uint32_t op) { return (0x181u >> op) & 1u; }     // M = X

struct KParams {
    int32_t min_quality, window, do_trim, do_count;
    int32_t ref_len, max_primer_len;
    const int32_t *min_start, *max_end;  // -1 = None; valid when do_trim
    uint32_t epoch = 0;                  // number of the launch (a fast kernel that hands reads to the general pass leaves it in ctr[29]: the pass
                                         // after a launch without such reads ends at its first instruction instead of adding up 256 list lengths)
};

// Device pointers of amp_trim_out (any may be null except new_cig, which the launcher always provides).
struct DevOut {
    int32_t *new_pos;
    uint32_t *new_ncig;
    uint32_t *new_cig;
    int32_t *ref_len;
    uint8_t *trim_flags;
    uint8_t *status;
};

// Where insertion events go: the event list (bounded by cap; the cursor keeps counting so the
// host can detect a short reservation) and the per-position event tally used by calling.
constexpr int EV_SHARDS = 8;     // event cursors: ctr[16 .. 16+EV_SHARDS)
struct EventBuf {
    amp_ins_event *ev;         // EV_SHARDS regions of `cap` events each
    unsigned long long *ctr;   // [1] bound, [2] error reads, [3] deferred reads, [16+s] events of shard s
    uint32_t *ins_at;          // [ref_len]
    long long cap;             // per shard
    // One cursor per shard (block id modulo EV_SHARDS): a single hot counter serialises the chip.
    __device__ void record(int32_t pos, uint32_t read, int32_t lo, int32_t hi) const {
        const unsigned s = blockIdx.x & (EV_SHARDS - 1);
        unsigned long long idx = atomicAdd(&ctr[16 + s], 1ull);
        if ((long long)idx < cap) ev[(size_t)s * (size_t)cap + idx] = amp_ins_event{pos, read, lo, hi};
        atomicAdd(&ins_at[pos], 1u);
    }
};

// A CIGAR held as BAM words.  Every function below is generic over the buffer class CB, which
// provides get(i) / set(i, v) and a pointer member p: CigBuf<1> is contiguous global memory;
// the tile kernel uses an LDS-address-space buffer with one column per lane.
template <int S>
struct CigBuf {
    uint32_t *p;
    AMP_HD uint32_t get(int i) const { return p[i * S]; }
    AMP_HD void set(int i, uint32_t v) const { p[i * S] = v; }
};

template <class CB>
AMP_HD void reverse_ops(const CB &b, int n) {
    for (int i = 0, j = n - 1; i < j; ++i, --j) {
        uint32_t t = b.get(i);
        b.set(i, b.get(j));
        b.set(j, t);
    }
}

// Streams ops into a buffer, folding runs of the same op (the effect of fix_cigar, A:415-423).
template <class CB>
struct Emitter {
    CB b;
    int n = 0;
    uint32_t pend = 0;
    bool has = false;
    AMP_HD void push(uint32_t op, uint32_t len) {
        if (has && (pend & 15u) == op) {
            pend += len << 4;
        } else {
            if (has) b.set(n++, pend);
            pend = (len << 4) | op;
            has = true;
        }
    }
    AMP_HD uint32_t last_op() const { return pend & 15u; }
    AMP_HD int finish() {
        if (has) b.set(n++, pend);
        has = false;
        return n;
    }
};

// A:389-412 get_pos_on_query
template <class CB>
AMP_HD int32_t pos_on_query(const CB &c, int n, int32_t ref_pos, int32_t ref_start, int &err) {
    int32_t query_pos = 0, cur = ref_start;
    for (int i = 0; i < n; ++i) {
        uint32_t v = c.get(i), op = v & 15u;
        int32_t len = (int32_t)(v >> 4);
        if (op >= 9) { err = AMP_RS_CIGAR_OP; return 0; }
        if (consumes_ref(op)) {
            if (ref_pos <= cur + len) return consumes_query(op) ? query_pos + (ref_pos - cur) : query_pos;
            cur += len;
        }
        if (consumes_query(op)) query_pos += len;
    }
    return query_pos;
}

// A:363-386 get_pos_on_ref
template <class CB>
AMP_HD int32_t pos_on_ref(const CB &c, int n, int32_t query_pos, int32_t ref_start, int &err) {
    int32_t cur = 0, ref_pos = ref_start;
    for (int i = 0; i < n; ++i) {
        uint32_t v = c.get(i), op = v & 15u;
        int32_t len = (int32_t)(v >> 4);
        if (op >= 9) { err = AMP_RS_CIGAR_OP; return 0; }
        if (consumes_query(op)) {
            if (query_pos <= cur + len) return consumes_ref(op) ? ref_pos + (query_pos - cur) : ref_pos;
            cur += len;
        }
        if (consumes_ref(op)) ref_pos += len;
    }
    return ref_pos;
}

// pysam accessors
template <class CB>
AMP_HD int32_t reference_length(const CB &c, int n) {
    int32_t r = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t v = c.get(i), op = v & 15u;
        if (op < 9 && consumes_ref(op)) r += (int32_t)(v >> 4);
    }
    return r ? r : 1;
}
template <class CB>
AMP_HD int32_t query_alignment_start(const CB &c, int n, int32_t lseq, int &err) {
    int32_t off = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t v = c.get(i), op = v & 15u;
        if (op == OP_H) {
            if (off != 0 && off != lseq) { err = AMP_RS_CLIP; return 0; }
        } else if (op == OP_S) {
            off += (int32_t)(v >> 4);
        } else {
            break;
        }
    }
    return off;
}
template <class CB>
AMP_HD int32_t query_alignment_end(const CB &c, int n, int32_t lseq, int &err) {
    int32_t end = lseq;
    if (end == 0) {
        for (int i = 0; i < n; ++i) {
            uint32_t v = c.get(i), op = v & 15u;
            if (op == OP_M || op == OP_I || op == OP_EQ || op == OP_X || (op == OP_S && end == 0)) end += (int32_t)(v >> 4);
        }
        return end;
    }
    for (int k = n - 1; k >= 1; --k) {  // element 0 is never examined
        uint32_t v = c.get(k), op = v & 15u;
        if (op == OP_H) {
            if (end != lseq) { err = AMP_RS_CLIP; return 0; }
        } else if (op == OP_S) {
            end -= (int32_t)(v >> 4);
        } else {
            break;
        }
    }
    return end;
}
// Python slice bounds of seq[a:b] for a sequence of length L
AMP_HD void py_slice(int32_t a, int32_t b, int32_t L, int32_t &lo, int32_t &hi) {
    if (a < 0) { a += L; if (a < 0) a = 0; } else if (a > L) a = L;
    if (b < 0) { b += L; if (b < 0) b = 0; } else if (b > L) b = L;
    if (b < a) b = a;
    lo = a; hi = b;
}

// Primer clip from the front of `src` (A:467-510; the end clip A:524-555 runs it over the
// reversed CIGAR without position tracking).  Returns the reference advance.
template <bool TRACK, class CB>
AMP_HD int32_t primer_clip(const CB &src, int n, bool reversed, int32_t del, Emitter<CB> &e, int &err) {
    bool pos_start = false;
    int32_t start_pos = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t v = src.get(reversed ? n - 1 - i : i), op = v & 15u;
        int32_t len = (int32_t)(v >> 4);
        if (del == 0 && pos_start) { e.push(op, len); continue; }
        if (op >= 9) { err = AMP_RS_CIGAR_OP; return 0; }
        if (del == 0 && consumes_query(op) && consumes_ref(op)) { pos_start = true; e.push(op, len); continue; }
        int32_t ref_add = 0;
        if (consumes_query(op)) {
            if (del >= len) e.push(OP_S, len);
            else if (del > 0) e.push(OP_S, del);
            else { e.push(OP_S, len); continue; }
            ref_add = del < len ? del : len;
            int32_t rest = len - del > 0 ? len - del : 0;
            del = del - len > 0 ? del - len : 0;
            if (rest > 0) e.push(op, rest);
            uint32_t last = e.last_op();
            if (del == 0 && consumes_query(last) && consumes_ref(last)) pos_start = true;
        } else if (consumes_ref(op)) {
            ref_add = len;
        }
        if (TRACK && consumes_ref(op)) start_pos += ref_add;
    }
    return start_pos;
}

// Quality clip from the front of `src` (A:597-622; A:658-683 over the reversed CIGAR).
template <class CB>
AMP_HD void quality_clip(const CB &src, int n, bool reversed, int32_t del, Emitter<CB> &e, int &err) {
    for (int i = 0; i < n; ++i) {
        uint32_t v = src.get(reversed ? n - 1 - i : i), op = v & 15u;
        int32_t len = (int32_t)(v >> 4);
        if (del == 0 || op == OP_S || op == OP_H) { e.push(op, len); continue; }
        if (op >= 9) { err = AMP_RS_CIGAR_OP; return; }
        if (consumes_query(op)) {
            e.push(OP_S, del >= len ? len : del);
            int32_t rest = len - del > 0 ? len - del : 0;
            del = del - len > 0 ? del - len : 0;
            if (rest > 0) e.push(op, rest);
        }
    }
}

// First index (from the 3' side) at which the sliding-window mean drops below min_quality:
// A:566-587 (reverse strand, returns i = number of leading bases to clip) and A:630-649
// (forward strand, returns i = number of leading bases kept).  total/window < q is evaluated
// as total < q*window, which is exact for integers.
// The bytes come through two 8-byte caches (one per edge of the window): on the GPU this loop runs on one lane, and a
// byte load per step -- a memory round trip each, the next step waits for it -- made the scan of a 400-base read take
// half a millisecond.
struct QualBytes8 {
    const uint8_t *base;           // 8-byte aligned; index 0 of the scan sits at byte `shift`
    int64_t shift;
    int64_t blk = -1;
    uint64_t w = 0;
    AMP_HD uint32_t at(int32_t k) {
        const int64_t j = (int64_t)k + shift;
        if ((j >> 3) != blk) {
            blk = j >> 3;
            const uint32_t *p = (const uint32_t *)(base + blk * 8);
            w = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
        }
        return (uint32_t)(w >> ((j & 7) * 8)) & 0xFFu;
    }
};
AMP_HD int32_t quality_scan(const uint8_t *q, int32_t qlen, int32_t width, int32_t min_quality, bool reverse) {
    int64_t total = 0, mq = min_quality;
    int32_t window = width < qlen ? width : qlen;
    const int64_t mis = (int64_t)((uintptr_t)q & 7u);
    QualBytes8 lead{q - mis, mis}, trail{q - mis, mis};
    if (reverse) {
        int32_t i = qlen;
        for (int32_t off = 1; off < window; ++off) total += lead.at(i - off);
        while (i > 0) {
            if (window > i) window -= 1; else total += lead.at(i - window);
            if (total < mq * window) break;
            total -= trail.at(i - 1);
            i -= 1;
        }
        return i;
    }
    int32_t i = 0;
    for (int32_t off = 0; off < window - 1; ++off) total += lead.at(off);
    while (i < qlen) {
        if (qlen - window < i) window -= 1; else total += lead.at(i + window - 1);
        if (total < mq * window) break;
        total -= trail.at(i);
        i += 1;
    }
    return i;
}

struct TrimState {
    int32_t pos;      // reference_start (updated by the start clip, A:514)
    int n;            // ops in `cur`
    uint32_t flags;   // AMP_TRIM_* bits
    int err;          // amp_read_status
};

// Stage 1+2 of trim_read: primer clips (A:450-558).  `cur` holds the CIGAR, `tmp` is scratch
// of the same capacity (n + 3); on return `cur` is the buffer holding the result.
template <class CB>
AMP_HD void trim_primers(const KParams &P, TrimState &st, uint32_t flag, int32_t tlen, int32_t lseq,
                         CB &cur, CB &tmp) {
    const bool is_paired = flag & 1u, is_reverse = (flag & 0x10u) != 0;
    const int32_t rs = st.pos;
    if ((uint32_t)rs >= (uint32_t)P.ref_len) { st.err = AMP_RS_INDEX_REF; return; }       // A:450
    const int32_t re1 = rs + reference_length(cur, st.n) - 1;
    if ((uint32_t)re1 >= (uint32_t)P.ref_len) { st.err = AMP_RS_INDEX_REF; return; }      // A:451
    const int32_t left_max_end = P.max_end[rs];
    const int32_t right_min_start = P.min_start[re1];
    const int32_t at = tlen < 0 ? -tlen : tlen;
    const bool isize_flag = ((int64_t)at - P.max_primer_len) > (int64_t)lseq;              // A:452
    if (!(is_paired && isize_flag && is_reverse) && left_max_end >= 0) {                   // A:460
        st.flags |= AMP_TRIM_PRIMER_START;
        int32_t del = pos_on_query(cur, st.n, left_max_end + 1, st.pos, st.err);           // A:463
        if (st.err) return;
        Emitter<CB> e{tmp};
        int32_t adv = primer_clip<true>(cur, st.n, false, del, e, st.err);
        if (st.err) return;
        st.n = e.finish();
        st.pos += adv;                                                                     // A:514
        CB t = cur; cur = tmp; tmp = t;
    }
    if (!(is_paired && isize_flag && !is_reverse) && right_min_start >= 0) {               // A:517
        st.flags |= AMP_TRIM_PRIMER_END;
        int32_t del = lseq - pos_on_query(cur, st.n, right_min_start, st.pos, st.err);     // A:520
        if (st.err) return;
        Emitter<CB> e{tmp};
        primer_clip<false>(cur, st.n, true, del, e, st.err);
        if (st.err) return;
        st.n = e.finish();
        reverse_ops(tmp, st.n);                                                            // A:558
        CB t = cur; cur = tmp; tmp = t;
    }
}

// Stage 3 of trim_read given the scan result `i` of quality_scan (A:589-625, A:651-686).
template <class CB>
AMP_HD void trim_quality_apply(TrimState &st, bool is_reverse, int32_t i, int32_t qlen, int32_t qs,
                               CB &cur, CB &tmp) {
    if (is_reverse) {
        const int32_t del = i;
        int32_t start_pos = pos_on_ref(cur, st.n, del + qs - 1, st.pos, st.err);           // A:591
        if (st.err) return;
        if (start_pos > st.pos) {                                                          // A:594
            st.flags |= AMP_TRIM_QUALITY;
            Emitter<CB> e{tmp};
            quality_clip(cur, st.n, false, del, e, st.err);
            if (st.err) return;
            st.n = e.finish();                              // reference_start is NOT advanced
            CB t = cur; cur = tmp; tmp = t;
        }
    } else {
        const int32_t del = qlen - i;
        (void)pos_on_ref(cur, st.n, del, st.pos, st.err);                                  // A:653 (may raise)
        if (st.err) return;
        if (del != 0) {                                                                    // A:656
            st.flags |= AMP_TRIM_QUALITY;
            Emitter<CB> e{tmp};
            quality_clip(cur, st.n, true, del, e, st.err);
            if (st.err) return;
            st.n = e.finish();
            reverse_ops(tmp, st.n);                                                        // A:686
            CB t = cur; cur = tmp; tmp = t;
        }
    }
}

// Aligned-quality window [lo, lo+qlen) of A:561 for the current CIGAR; sets err like the
// pysam accessors do.  Returns false when the read cannot be quality-trimmed.
template <class CB>
AMP_HD bool quality_window(TrimState &st, int32_t lseq, bool have_qual, const CB &cur,
                           int32_t &qs, int32_t &lo, int32_t &qlen) {
    if (lseq == 0) { st.err = AMP_RS_NO_QUAL; return false; }
    qs = query_alignment_start(cur, st.n, lseq, st.err);
    if (st.err) return false;
    int32_t qe = query_alignment_end(cur, st.n, lseq, st.err);
    if (st.err) return false;
    if (!have_qual) { st.err = AMP_RS_NO_QUAL; return false; }
    int32_t hi;
    py_slice(qs, qe, lseq, lo, hi);
    qlen = hi - lo;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Closed forms of the trims for the commonest read: ONE match op covering the whole query
// ("150M").  The loops above reduce to a few comparisons; the result is always [S a][op m][S c]
// (zero-length parts absent; m == 0 collapses to [S L]).  Derived case by case from primer_clip /
// quality_clip above and checked against them on the golden vectors (tests/hostsim runs both).
// ---------------------------------------------------------------------------------------------
struct SimpleCig {
    uint32_t op;          // M, = or X
    int32_t a, m, c;      // leading soft clip, match length, trailing soft clip; a + m + c == l_seq
    template <class CB>
    AMP_HD int store(const CB &b) const {
        int n = 0;
        if (a > 0) b.set(n++, ((uint32_t)a << 4) | OP_S);
        if (m > 0) b.set(n++, ((uint32_t)m << 4) | op);
        if (c > 0) b.set(n++, ((uint32_t)c << 4) | OP_S);
        return n;
    }
};

AMP_HD bool is_simple_cigar(int n, uint32_t w0, int32_t lseq) {
    const uint32_t op = w0 & 15u;
    return n == 1 && (op == OP_M || op == OP_EQ || op == OP_X) && lseq > 0 && (int32_t)(w0 >> 4) == lseq;
}

// trim_primers for a simple read given the two table entries of A:450-451; `sc` starts as {op, 0, l_seq, 0}
AMP_HD void trim_primers_simple_tab(const KParams &P, TrimState &st, uint32_t flag, int32_t tlen, int32_t lseq, SimpleCig &sc,
                                    int32_t left_max_end, int32_t right_min_start) {
    const bool is_paired = flag & 1u, is_reverse = (flag & 0x10u) != 0;
    const int32_t rs = st.pos, L = lseq;
    const int32_t at = tlen < 0 ? -tlen : tlen;
    const bool isize_flag = ((int64_t)at - P.max_primer_len) > (int64_t)lseq;              // A:452
    if (!(is_paired && isize_flag && is_reverse) && left_max_end >= 0) {                   // A:460
        st.flags |= AMP_TRIM_PRIMER_START;
        // pos_on_query([op L], left_max_end + 1): inside the op when ref_pos <= rs + L, else the query length
        const int64_t rp = (int64_t)left_max_end + 1;
        const int32_t del = rp <= (int64_t)rs + L ? (int32_t)(rp - rs) : L;
        if (del != 0) {
            if (del < 0) { sc.a = L; sc.m = 0; }                               // A:483-485: whole op -> S, start unchanged
            else if (del >= L) { sc.a = L; sc.m = 0; st.pos += L; }
            else { sc.a = del; sc.m = L - del; st.pos += del; }
        }
    }
    if (!(is_paired && isize_flag && !is_reverse) && right_min_start >= 0) {               // A:517
        st.flags |= AMP_TRIM_PRIMER_END;
        if (sc.m > 0) {
            const int64_t qp = (int64_t)right_min_start <= (int64_t)st.pos + sc.m ? (int64_t)sc.a + ((int64_t)right_min_start - st.pos) : (int64_t)L;
            const int64_t del = (int64_t)L - qp;                               // >= 0; may exceed the match length
            if (del > 0) {
                if (del >= sc.m) { sc.a = L; sc.m = 0; }                       // everything soft-clipped
                else { sc.c = (int32_t)del; sc.m -= (int32_t)del; }
            }
        }
    }
}

// trim_primers for a simple read
AMP_HD void trim_primers_simple(const KParams &P, TrimState &st, uint32_t flag, int32_t tlen, int32_t lseq, SimpleCig &sc) {
    const int32_t rs = st.pos;
    if ((uint32_t)rs >= (uint32_t)P.ref_len) { st.err = AMP_RS_INDEX_REF; return; }       // A:450
    const int32_t re1 = rs + lseq - 1;
    if ((uint32_t)re1 >= (uint32_t)P.ref_len) { st.err = AMP_RS_INDEX_REF; return; }      // A:451
    trim_primers_simple_tab(P, st, flag, tlen, lseq, sc, P.max_end[rs], P.min_start[re1]);
}

// trim_quality_apply for a simple read (the window of quality_window is lo = a, qlen = m)
AMP_HD void trim_quality_apply_simple(TrimState &st, bool is_reverse, int32_t i, int32_t qlen, SimpleCig &sc) {
    if (sc.m <= 0) return;
    if (is_reverse) {
        const int32_t del = i;                     // pos_on_ref(del + qs - 1) > pos  <=>  del >= 2  (A:591-594)
        if (del >= 2) {
            st.flags |= AMP_TRIM_QUALITY;
            sc.a += del; sc.m -= del;              // reference_start is NOT advanced
            if (sc.m == 0) { sc.a += sc.c; sc.c = 0; }
        }
    } else {
        const int32_t del = qlen - i;
        if (del != 0) {                                                                    // A:656
            st.flags |= AMP_TRIM_QUALITY;
            sc.c += del; sc.m -= del;
            if (sc.m == 0) { sc.a += sc.c; sc.c = 0; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Closed forms for reads with at most ONE insertion or deletion:  [S a][op m1][X k][op m2][S c]
// with X = I or D (kind 1 / 2; kind 0: no indel, k = m2 = 0).  A clip that stops inside or right
// in front of the indel leaves it directly behind the soft clip ([S a][X k][op m2]: m1 == 0, or
// m2 == 0 when it came from the other end); the forms below take such shapes too.  What they do
// not cover -- a second clip that reaches an indel which already touches a clip -- sets `punt`
// and the generic code does the read.
// Derived rule by rule from primer_clip / quality_clip above; tests/hostsim fuzzes them against
// trim_read_serial (tests/test_hostsim_golden.py::test_two_segment_closed_forms).
// ---------------------------------------------------------------------------------------------
struct Cig2 {
    uint32_t op;                  // M, = or X: the op of both match segments
    int32_t a, m1, k, m2, c;
    int32_t kind;                 // 0 none, 1 insertion, 2 deletion
    bool punt;
    AMP_HD int32_t query_len() const { return a + m1 + (kind == 1 ? k : 0) + m2 + c; }
    AMP_HD int32_t ref_len() const { const int32_t r = m1 + m2 + (kind == 2 ? k : 0); return r ? r : 1; }
    AMP_HD void canon() {         // no match left: one soft clip
        if (kind == 0 && m1 == 0) { a += c; c = 0; }
    }
    template <class CB>
    AMP_HD int store(const CB &b) const {
        int n = 0;
        if (a > 0) b.set(n++, ((uint32_t)a << 4) | OP_S);
        if (m1 > 0) b.set(n++, ((uint32_t)m1 << 4) | op);
        if (kind) { b.set(n++, ((uint32_t)k << 4) | (kind == 1 ? OP_I : OP_D)); if (m2 > 0) b.set(n++, ((uint32_t)m2 << 4) | op); }
        if (c > 0) b.set(n++, ((uint32_t)c << 4) | OP_S);
        return n;
    }
    AMP_HD void mirror() { int32_t t = a; a = c; c = t; if (kind) { t = m1; m1 = m2; m2 = t; } }
};

// the shape of an input CIGAR of n ops (first three words given), or false
AMP_HD bool cig2_from_words(int n, uint32_t w0, uint32_t w1, uint32_t w2, int32_t lseq, Cig2 &s) {
    const uint32_t o0 = w0 & 15u;
    if (!(o0 == OP_M || o0 == OP_EQ || o0 == OP_X) || lseq <= 0) return false;
    s.op = o0; s.a = 0; s.c = 0; s.punt = false;
    if (n == 1) { s.m1 = (int32_t)(w0 >> 4); s.kind = 0; s.k = 0; s.m2 = 0; return s.m1 == lseq; }
    if (n != 3) return false;
    const uint32_t o1 = w1 & 15u, o2 = w2 & 15u;
    if (o2 != o0 || !(o1 == OP_I || o1 == OP_D)) return false;
    s.m1 = (int32_t)(w0 >> 4); s.k = (int32_t)(w1 >> 4); s.m2 = (int32_t)(w2 >> 4); s.kind = o1 == OP_I ? 1 : 2;
    if (s.m1 <= 0 || s.k <= 0 || s.m2 <= 0) return false;
    return s.query_len() == lseq;
}

// the same for input CIGARs that carry soft clips at their ends (aligners clip adapters and primers): up to five ops
// [S a][op m1]([I|D k][op m2])[S c]
AMP_HD bool cig2_from_words5(int n, const uint32_t (&w)[5], int32_t lseq, Cig2 &s) {
    // (no indexing by a run-time value: on the GPU that would put the five words into scratch memory)
    if (n < 1 || n > 5 || lseq <= 0) return false;
    s.a = 0; s.c = 0; s.k = 0; s.m2 = 0; s.kind = 0; s.punt = false;
    const bool lead = (w[0] & 15u) == OP_S;
    if (lead) s.a = (int32_t)(w[0] >> 4);
    const uint32_t v0 = lead ? w[1] : w[0], v1 = lead ? w[2] : w[1], v2 = lead ? w[3] : w[2], v3 = lead ? w[4] : w[3];
    const int m = n - (lead ? 1 : 0);                      // ops behind the leading clip: M | M S | M X M | M X M S
    if (lead && s.a <= 0) return false;
    if (m < 1 || m > 4) return false;
    const uint32_t o0 = v0 & 15u;
    if (!(o0 == OP_M || o0 == OP_EQ || o0 == OP_X)) return false;
    s.op = o0; s.m1 = (int32_t)(v0 >> 4);
    if (s.m1 <= 0) return false;
    uint32_t tail = 0;                                     // the trailing clip's word, if any
    bool has_tail = false;
    if (m >= 3) {
        const uint32_t o1 = v1 & 15u;
        if (!(o1 == OP_I || o1 == OP_D) || (v2 & 15u) != o0) return false;
        s.kind = o1 == OP_I ? 1 : 2; s.k = (int32_t)(v1 >> 4); s.m2 = (int32_t)(v2 >> 4);
        if (s.k <= 0 || s.m2 <= 0) return false;
        has_tail = m == 4; tail = v3;
    } else {
        has_tail = m == 2; tail = v1;
    }
    if (has_tail) {
        if ((tail & 15u) != OP_S) return false;
        s.c = (int32_t)(tail >> 4);
        if (s.c <= 0) return false;
    }
    return s.query_len() == lseq;
}

// get_pos_on_query (A:389-412) on the shape
AMP_HD int32_t cig2_pos_on_query(const Cig2 &s, int64_t ref_pos, int64_t ref_start) {
    int64_t q = s.a, cur = ref_start;
    if (s.m1 > 0) { if (ref_pos <= cur + s.m1) return (int32_t)(q + (ref_pos - cur)); cur += s.m1; q += s.m1; }
    if (s.kind == 1) q += s.k;
    else if (s.kind == 2) { if (ref_pos <= cur + s.k) return (int32_t)q; cur += s.k; }
    if (s.kind && s.m2 > 0) { if (ref_pos <= cur + s.m2) return (int32_t)(q + (ref_pos - cur)); q += s.m2; }
    return (int32_t)(q + s.c);
}

// get_pos_on_ref (A:363-386) on the shape
AMP_HD int32_t cig2_pos_on_ref(const Cig2 &s, int64_t query_pos, int64_t ref_start) {
    int64_t cur = 0, r = ref_start;
    if (s.a > 0) { if (query_pos <= cur + s.a) return (int32_t)r; cur += s.a; }
    if (s.m1 > 0) { if (query_pos <= cur + s.m1) return (int32_t)(r + (query_pos - cur)); cur += s.m1; r += s.m1; }
    if (s.kind == 1) { if (query_pos <= cur + s.k) return (int32_t)r; cur += s.k; }
    else if (s.kind == 2) r += s.k;
    if (s.kind && s.m2 > 0) { if (query_pos <= cur + s.m2) return (int32_t)(r + (query_pos - cur)); cur += s.m2; r += s.m2; }
    return (int32_t)r;   // (a trailing soft clip returns r either way)
}

// primer clip of `del` query bases from the front (A:467-510); returns the reference advance
AMP_HD int32_t cig2_primer_clip(Cig2 &s, int32_t del) {
    if (del == 0) return 0;
    if (del < 0) {                // every query-consuming op falls into the "else" arm: all soft clip; a deletion still advances
        const int32_t adv = s.kind == 2 ? s.k : 0;
        s.a = s.query_len(); s.m1 = s.m2 = s.k = s.c = 0; s.kind = 0;
        return adv;
    }
    int32_t adv = 0, A = s.a;
    if (s.a > 0) del = del >= s.a ? del - s.a : 0;                    // a soft clip stays one and eats its share
    if (s.m1 > 0) {
        if (del == 0) return 0;                                       // first match op starts the alignment: rest copied
        if (del < s.m1) { s.a = A + del; s.m1 -= del; return del; }
        A += s.m1; adv += s.m1; del -= s.m1; s.m1 = 0;
    }
    if (s.kind && s.m2 == 0) { s.punt = true; return 0; }               // an indel that already touches the far clip
    if (s.kind == 1) {
        if (del > 0 && del < s.k) {                                    // insertion cut in two: its rest stays, behind the clip
            s.a = A + del; s.m1 = 0; s.k -= del;
            return adv;
        }
        A += s.k; del = del >= s.k ? del - s.k : 0;                    // (del == 0: not started yet -> soft clip as well)
    } else if (s.kind == 2) {
        adv += s.k;                                                    // dropped, the start jumps over it
    }
    int32_t m = s.kind ? s.m2 : 0;
    if (m > 0 && del > 0) {
        if (del >= m) { A += m; adv += m; m = 0; }
        else { A += del; adv += del; m -= del; }
    }
    s.a = A; s.m1 = m; s.kind = 0; s.k = 0; s.m2 = 0;
    s.canon();
    return adv;
}

// quality clip of `del` aligned bases from the front (A:597-622)
AMP_HD void cig2_quality_clip(Cig2 &s, int32_t del) {
    if (del == 0) return;
    int32_t A = s.a;
    if (s.m1 > 0) {
        if (del < s.m1) { s.a = A + del; s.m1 -= del; return; }
        A += s.m1; del -= s.m1; s.m1 = 0;
    }
    if (s.kind && s.m2 == 0) { s.punt = true; return; }                 // an indel that already touches the far clip
    if (s.kind == 1) {
        if (del < s.k) { s.a = A + del; s.m1 = 0; s.k -= del; return; }  // del == 0: the insertion is copied behind the clip; else its rest
        A += s.k; del -= s.k;
    } else if (s.kind == 2) {
        if (del == 0) { s.a = A; s.m1 = 0; return; }                   // the deletion is copied behind the clip
    }
    int32_t m = s.kind ? s.m2 : 0;
    if (m > 0 && del > 0) {
        if (del >= m) { A += m; m = 0; }
        else { A += del; m -= del; }
    }
    s.a = A; s.m1 = m; s.kind = 0; s.k = 0; s.m2 = 0;
    s.canon();
}

// Stage 1+2 of trim_read (A:450-558) on the shape, given the two table entries and the outcome of the template-length
// test of A:452
AMP_HD void cig2_trim_primers_isize(TrimState &st, uint32_t flag, bool isize_flag, int32_t lseq, Cig2 &s,
                                    int32_t left_max_end, int32_t right_min_start) {
    const bool is_paired = flag & 1u, is_reverse = (flag & 0x10u) != 0;
    if (!(is_paired && isize_flag && is_reverse) && left_max_end >= 0) {                   // A:460
        st.flags |= AMP_TRIM_PRIMER_START;
        const int32_t del = cig2_pos_on_query(s, (int64_t)left_max_end + 1, st.pos);       // A:463
        st.pos += cig2_primer_clip(s, del);                                                // A:514
        if (s.punt) return;
    }
    if (!(is_paired && isize_flag && !is_reverse) && right_min_start >= 0) {               // A:517
        st.flags |= AMP_TRIM_PRIMER_END;
        const int32_t del = lseq - cig2_pos_on_query(s, right_min_start, st.pos);          // A:520
        s.mirror();
        (void)cig2_primer_clip(s, del);
        s.mirror();
        s.canon();
    }
}
AMP_HD void cig2_trim_primers(const KParams &P, TrimState &st, uint32_t flag, int32_t tlen, int32_t lseq, Cig2 &s,
                              int32_t left_max_end, int32_t right_min_start) {
    const int32_t at = tlen < 0 ? -tlen : tlen;
    cig2_trim_primers_isize(st, flag, ((int64_t)at - P.max_primer_len) > (int64_t)lseq, lseq, s, left_max_end, right_min_start);   // A:452
}

// aligned-quality window of the shape (quality_window above): [lo, lo + qlen)
AMP_HD void cig2_quality_window(const Cig2 &s, int32_t lseq, int32_t &lo, int32_t &qlen) {
    const bool single = s.m1 == 0 && s.kind == 0;           // one soft clip covering the read: its end is not examined
    const int32_t qs = s.a, qe = single ? lseq : lseq - s.c;
    lo = qs > lseq ? lseq : qs;
    const int32_t hi = qe < lo ? lo : qe;
    qlen = hi - lo;
}

// Stage 3 of trim_read given the scan result i (A:589-625, A:651-686)
AMP_HD void cig2_trim_quality(TrimState &st, bool is_reverse, int32_t i, int32_t qlen, Cig2 &s) {
    if (is_reverse) {
        const int32_t del = i;
        if (cig2_pos_on_ref(s, (int64_t)del + s.a - 1, st.pos) > st.pos) {                 // A:591-594
            st.flags |= AMP_TRIM_QUALITY;
            cig2_quality_clip(s, del);                      // reference_start is NOT advanced
        }
    } else {
        const int32_t del = qlen - i;
        if (del != 0) {                                                                    // A:656
            st.flags |= AMP_TRIM_QUALITY;
            s.mirror();
            cig2_quality_clip(s, del);
            s.mirror();
            s.canon();
        }
    }
}

// Whole trim_read (A:426-687) for one read, scanning qualities serially.
template <class CB>
AMP_HD void trim_read_serial(const KParams &P, TrimState &st, uint32_t flag, int32_t tlen, int32_t lseq,
                             const uint8_t *qual, bool have_qual, CB &cur, CB &tmp) {
    const bool rev = (flag & 0x10u) != 0;
    if (st.n == 1 && is_simple_cigar(1, cur.get(0), lseq)) {
        SimpleCig sc{cur.get(0) & 15u, 0, lseq, 0};
        trim_primers_simple(P, st, flag, tlen, lseq, sc);
        if (st.err) return;
        st.n = sc.store(cur);
        if (!have_qual) { st.err = AMP_RS_NO_QUAL; return; }
        const int32_t lo = sc.m > 0 ? sc.a : lseq, qlen = sc.m;
        const int32_t i = quality_scan(qual + lo, qlen, P.window, P.min_quality, rev);
        trim_quality_apply_simple(st, rev, i, qlen, sc);
        st.n = sc.store(cur);
        return;
    }
    trim_primers(P, st, flag, tlen, lseq, cur, tmp);
    if (st.err) return;
    int32_t qs, lo, qlen;
    if (!quality_window(st, lseq, have_qual, cur, qs, lo, qlen)) return;
    int32_t i = quality_scan(qual + lo, qlen, P.window, P.min_quality, rev);
    trim_quality_apply(st, rev, i, qlen, qs, cur, tmp);
}

// BAM 4-bit code -> count-table column; 0xFF = KeyError (A:892 has keys A C G T N '-')
AMP_HD uint32_t code_to_col(uint32_t code) {
    // codes 1,2,4,8,15 -> 0,1,2,3,4
    return code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : code == 8 ? 3u : code == 15 ? 4u : 0xFFu;
}
AMP_HD uint32_t base_code(const uint8_t *seq, int64_t base_off, int32_t q) {
    int64_t k = base_off + q;
    uint32_t b = seq[k >> 1];
    return (k & 1) ? (b & 15u) : (b >> 4);
}

// Iterator over get_aligned_pairs() without materialising the list.
template <class CB>
struct PairIter {
    CB c;
    int n, k;
    int32_t j, len, q, r;
    uint32_t op;
    AMP_HD void init(const CB &cig, int nops, int32_t ref_start) {
        c = cig; n = nops; k = -1; j = 0; len = 0; q = 0; r = ref_start; op = OP_H;
    }
    // fetches the next pair; pq/pr = -1 for None
    AMP_HD bool next(int32_t &pq, int32_t &pr) {
        while (j >= len) {
            if (++k >= n) return false;
            uint32_t v = c.get(k);
            op = v & 15u; j = 0;
            len = (op == OP_H || op >= 9) ? 0 : (int32_t)(v >> 4);
        }
        ++j;
        if (is_match_op(op)) { pq = q++; pr = r++; }
        else if (op == OP_D || op == OP_N) { pq = -1; pr = r++; }
        else { pq = q++; pr = -1; }  // I, S and (pysam quirk) P
        return true;
    }
};

// update_base_counts (A:690-753) for one read as an exact sequential walk.  Sink provides
//   void add(int32_t ref_pos, uint32_t col)   and   void event(int32_t ref_pos, int32_t from, int32_t to)
// Plain accessor of one read's qualities and 4-bit base codes.
struct ReadBytes {
    const uint8_t *seq; int64_t base_off; const uint8_t *q;
    AMP_HD uint32_t qual(int32_t i) { return q[i]; }
    AMP_HD uint32_t code(int32_t i) { return base_code(seq, base_off, i); }
};

// The same with an 8-base register cache: the walk visits query indices in ascending order, so one
// 8-byte + one 4-byte load serve eight bases (reads start on 8-base boundaries).
struct ReadBytesCached {
    const uint8_t *seq; int64_t base_off; const uint8_t *q;
    int32_t blk = -1;
    uint32_t q0 = 0, q1 = 0, s = 0;
    AMP_HD void fetch(int32_t i) {
        blk = i >> 3;
        const uint32_t *qp = (const uint32_t *)(q + (int64_t)blk * 8);
        q0 = qp[0]; q1 = qp[1];
        s = *(const uint32_t *)(seq + ((base_off + (int64_t)blk * 8) >> 1));
    }
    AMP_HD uint32_t qual(int32_t i) {
        if ((i >> 3) != blk) fetch(i);
        return (((i & 4) ? q1 : q0) >> ((i & 3) * 8)) & 0xFFu;
    }
    AMP_HD uint32_t code(int32_t i) {
        if ((i >> 3) != blk) fetch(i);
        const uint32_t byte = (s >> (((i & 7) >> 1) * 8)) & 0xFFu;
        return (i & 1) ? (byte & 15u) : (byte >> 4);
    }
};

template <class CB, class Sink, class RB>
AMP_HD int count_read_walk(const KParams &P, const CB &cig, int n, int32_t ref_start, int32_t lseq,
                           RB rb, bool have_qual, Sink &sink) {
    int err = 0;
    const int32_t qs = query_alignment_start(cig, n, lseq, err);                           // A:700
    if (err) return err;
    const int32_t qe = query_alignment_end(cig, n, lseq, err);                             // A:701
    if (err) return err;
    if (lseq == 0) return AMP_RS_NO_SEQ;                                                   // A:702
    const int32_t ref_end = ref_start + reference_length(cig, n);                          // A:705
    const uint32_t G = (uint32_t)P.ref_len;
    const int32_t mq = P.min_quality;
    PairIter<CB> it;
    it.init(cig, n, ref_start);
    int32_t q, r;
    bool pending = false;  // a pair fetched by the insertion scan and handed back (A:743)
    int32_t pend_q = 0, pend_r = 0;
    for (;;) {
        if (pending) { q = pend_q; r = pend_r; pending = false; }
        else if (!it.next(q, r)) break;
        if (q < 0) {                                                                       // A:714-715
            if ((uint32_t)r >= G) return AMP_RS_INDEX_REF;
            sink.add(r, 5u);
            continue;
        }
        if (!have_qual) return AMP_RS_NO_QUAL;
        if (q >= lseq) return AMP_RS_INDEX_QUERY;
        if ((int32_t)rb.qual(q) < mq) continue;                                            // A:718
        if (q < qs) continue;                                                              // A:722
        if (q >= qe) break;                                                                // A:726
        if (r < 0) {                                                                       // A:730-748
            const int32_t q0 = q;
            bool q_none = false;
            while (r < 0 && !q_none && q < qe) {
                if (q >= lseq) return AMP_RS_INDEX_QUERY;
                if ((int32_t)rb.qual(q) < mq) break;
                if (!it.next(q, r)) return AMP_RS_INDEX_PAIRS;                             // A:734
                if (q < 0) q_none = true;
            }
            int32_t lo, hi;
            if (r == 0) {                                                                  // A:735-736
                if (q_none) return AMP_RS_TYPE;
                py_slice(q0, q + 1, lseq, lo, hi);
            } else if (q_none) {
                py_slice(q0 - 1, lseq, lseq, lo, hi);                                      // seq[a:None]
            } else {
                py_slice(q0 - 1, q, lseq, lo, hi);                                         // A:738
            }
            int32_t ins_pos;
            if (r < 0) ins_pos = ref_end;                                                  // A:739-740
            else { ins_pos = r; pending = true; pend_q = q; pend_r = r; }                  // A:742-743
            ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;                                   // A:744
            if ((uint32_t)ins_pos >= G) return AMP_RS_INDEX_REF;
            sink.event(ins_pos, lo, hi);
            continue;
        }
        if ((uint32_t)r >= G) return AMP_RS_INDEX_REF;                                     // A:751-753
        const uint32_t col = code_to_col(rb.code(q));
        if (col == 0xFFu) return AMP_RS_KEY_BASE;
        sink.add(r, col);
    }
    return 0;
}

// The indel effects of count_read_walk op by op, without the pair iterator (one step per BASE of every op there; here a step per op, a short loop
// over the bases of a deletion and over the bases of an insertion run): what the in-tile walk of k_tile runs.  A literal
// restatement of A:714-748 on a regular CIGAR -- the pairs of an insertion are (q, None), the run of good-quality bases
// from its first one ends at a low base or at the alignment end (the event is then counted at reference_end - 1 and that
// pair is swallowed), or at the pair behind the insertion: a match base (anchored there; the r == 0 quirk of A:735), a
// deleted base (q is None: the allele is SEQ[q0 - 1:]), a clipped base at the alignment end (counted at reference_end - 1)
// or nothing at all (A:734 raises).  I ops that follow each other are one run (zero-length ops and hard clips have no pairs).
template <class CB, class Sink, class QF>
AMP_HD int count_regular_ops(const KParams &P, const CB &cig, int n, int32_t ref_start, int32_t ref_end, int32_t lseq,
                                 int32_t qs, int32_t qe, const QF &qual, Sink &sink) {
    const uint32_t G = (uint32_t)P.ref_len;
    const int32_t mq = P.min_quality;
    int32_t q = 0, r = ref_start;
    int k = 0;
    while (k < n) {
        const uint32_t v = cig.get(k), op = v & 15u;
        const int32_t len = (int32_t)(v >> 4);
        ++k;
        if (is_match_op(op)) { q += len; r += len; continue; }
        if (op == OP_S) { q += len; continue; }
        if (op == OP_D || op == OP_N) {
            for (int32_t j = 0; j < len; ++j, ++r) {
                if ((uint32_t)r >= G) return AMP_RS_INDEX_REF;
                sink.add(r, 5u);
            }
            continue;
        }
        if (op != OP_I) continue;                     // hard clips
        int32_t L = len;
        uint32_t nop = OP_H;                          // the op whose first pair follows the run (OP_H: none)
        while (k < n) {
            const uint32_t w = cig.get(k), o2 = w & 15u;
            const int32_t l2 = (int32_t)(w >> 4);
            if (o2 == OP_H || l2 == 0) { ++k; continue; }
            if (o2 == OP_I) { L += l2; ++k; continue; }
            nop = o2;
            break;
        }
        const int32_t qi = q;
        q += L;
        int32_t j = 0;
        while (j < L) {
            if ((int32_t)qual(qi + j) < mq) { ++j; continue; }                             // A:718
            if (qi + j < qs) { ++j; continue; }                                            // A:722
            if (qi + j >= qe) return 0;                                                    // A:726
            const int32_t js = j;
            ++j;                                                                           // A:734: the next pair
            while (j < L && qi + j < qe && (int32_t)qual(qi + j) >= mq) ++j;
            int32_t lo, hi, ins_pos;
            if (j < L) {                              // cut short inside the run; that pair is swallowed
                py_slice(qi + js - 1, qi + j, lseq, lo, hi);                               // A:738
                ins_pos = ref_end; ++j;                                                    // A:739-740
            } else if (nop == OP_H) {
                return AMP_RS_INDEX_PAIRS;                                                 // A:734
            } else if (is_match_op(nop)) {
                if (r == 0) py_slice(qi + js, qi + L + 1, lseq, lo, hi);                   // A:735-736
                else py_slice(qi + js - 1, qi + L, lseq, lo, hi);                          // A:738
                ins_pos = r;                                                               // A:742
            } else if (nop == OP_D || nop == OP_N) {
                if (r == 0) return AMP_RS_TYPE;
                py_slice(qi + js - 1, lseq, lseq, lo, hi);                                 // seq[a:None]
                ins_pos = r;
            } else {                                  // a clipped base: of a regular read it sits at the alignment end
                if (qi + L < qe) return AMP_RS_INDEX_PAIRS;       // (not a regular read: the exact walk of the second pass decides)
                py_slice(qi + js - 1, qi + L, lseq, lo, hi);
                ins_pos = ref_end;
            }
            ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;                                   // A:744
            if ((uint32_t)ins_pos >= G) return AMP_RS_INDEX_REF;
            sink.event(ins_pos, lo, hi);
        }
    }
    return 0;
}

// update_base_counts (A:690-753) for the shape, part 1: the effects of its indel.  A deletion counts '-' at each of
// its positions (A:714-715).  An insertion gives one event per maximal run of good-quality inserted bases
// (A:730-748): a run that reaches the insertion's end is anchored on the base before
// it and counted in front of the next match base; a run cut short by a low-quality base is counted at
// reference_end - 1 and that base is swallowed.  qual(q) -> quality of query base q.
template <class Sink, class QF>
AMP_HD int cig2_indels(const KParams &P, const Cig2 &s, int32_t pos, int32_t lseq, const QF &qual, Sink &sink) {
    const uint32_t G = (uint32_t)P.ref_len;
    if (s.kind == 2) {
        for (int32_t j = 0; j < s.k; ++j) {
            const int32_t r = pos + s.m1 + j;
            if ((uint32_t)r >= G) return AMP_RS_INDEX_REF;
            sink.add(r, 5u);
        }
    } else if (s.kind == 1) {
        const int32_t q0 = s.a + s.m1, r2 = pos + s.m1, ref_end = pos + s.ref_len();
        int32_t j = 0;
        while (j < s.k) {
            if ((int32_t)qual(q0 + j) < P.min_quality) { ++j; continue; }                  // A:718
            const int32_t js = j;
            while (j < s.k && (int32_t)qual(q0 + j) >= P.min_quality) ++j;
            int32_t lo, hi, ins_pos;
            if (j == s.k && s.m2 > 0 && r2 == 0) py_slice(q0 + js, q0 + j + 1, lseq, lo, hi);   // A:735-736: the next match base sits on reference position 0
            else py_slice(q0 + js - 1, q0 + j, lseq, lo, hi);                              // A:738
            if (j == s.k) ins_pos = r2;                                                    // A:742
            else { ins_pos = ref_end; ++j; }                                               // A:739-740; the low base is consumed
            ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;                                   // A:744
            if ((uint32_t)ins_pos >= G) return AMP_RS_INDEX_REF;
            sink.event(ins_pos, lo, hi);
        }
    }
    return 0;
}

}  // namespace amp
