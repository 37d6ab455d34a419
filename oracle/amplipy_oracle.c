/*
 * amplipy_oracle.c -- CPU restatement of AmpliPy's trim + pileup + call path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (amplipy_amd/, libamplihip.so) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * It restates, function by function, the algorithm of /root/reference/AmpliPy.py v0.0.2
 * (cited as A:line) in scalar C over the same packed batch layout the GPU library takes
 * (include/amplihip.h).  It is deliberately literal -- the aligned pairs of a read are
 * materialised and walked exactly like A:706-753 -- and shares no code with the HIP
 * kernels, which use a different decomposition.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against
 * tests/golden files, which tools/make_golden.py produced by running the reference's own
 * functions.  The pysam accessor semantics (query_alignment_start/end, reference_length,
 * get_aligned_pairs) are not covered by any reference test and the
 * pysam source is not vendored: that part of the parity is UNPINNED.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/amplihip.h"

#define OP_M 0
#define OP_I 1
#define OP_D 2
#define OP_N 3
#define OP_S 4
#define OP_H 5
#define OP_P 6
#define OP_EQ 7
#define OP_X 8

/* A:43-44.  Indexing with op >= 9 is an IndexError in the reference. */
static const int CONSUME_QUERY[9] = {1, 1, 0, 0, 1, 0, 0, 1, 1};
static const int CONSUME_REF[9] = {1, 0, 1, 1, 0, 0, 0, 1, 1};

typedef struct { int32_t op; int64_t len; } cigop;

typedef struct {
    cigop *v;
    int n, cap;
} ciglist;

static int cl_reserve(ciglist *c, int cap) {
    if (cap <= c->cap) return 0;
    cigop *nv = (cigop *)realloc(c->v, (size_t)cap * sizeof(cigop));
    if (!nv) return -1;
    c->v = nv; c->cap = cap;
    return 0;
}
static int cl_push(ciglist *c, int32_t op, int64_t len) {
    if (c->n == c->cap && cl_reserve(c, c->cap ? 2 * c->cap : 16)) return -1;
    c->v[c->n].op = op; c->v[c->n].len = len; c->n++;
    return 0;
}
static void cl_reverse(ciglist *c) {
    for (int i = 0, j = c->n - 1; i < j; i++, j--) { cigop t = c->v[i]; c->v[i] = c->v[j]; c->v[j] = t; }
}

/* ---- A:415-423 fix_cigar: forward sweep, carrying the sum into the next element ---- */
static void fix_cigar(ciglist *c) {
    int out = 0;
    for (int i = 0; i < c->n; i++) {
        if (i < c->n - 1 && c->v[i].op == c->v[i + 1].op) { c->v[i + 1].len += c->v[i].len; continue; }
        c->v[out++] = c->v[i];
    }
    c->n = out;
}

/* ---- A:389-412 get_pos_on_query; *err set when a table is indexed with op >= 9 ---- */
static int64_t pos_on_query(const cigop *c, int n, int64_t ref_pos, int64_t ref_start, int *err) {
    int64_t query_pos = 0, cur_pos = ref_start;
    for (int i = 0; i < n; i++) {
        if (c[i].op >= 9) { *err = AMP_RS_CIGAR_OP; return 0; }
        if (CONSUME_REF[c[i].op]) {
            if (ref_pos <= cur_pos + c[i].len) {
                if (CONSUME_QUERY[c[i].op]) query_pos += ref_pos - cur_pos;
                return query_pos;
            }
            cur_pos += c[i].len;
        }
        if (CONSUME_QUERY[c[i].op]) query_pos += c[i].len;
    }
    return query_pos;
}

/* ---- A:363-386 get_pos_on_ref ---- */
static int64_t pos_on_ref(const cigop *c, int n, int64_t query_pos, int64_t ref_start, int *err) {
    int64_t cur_pos = 0, ref_pos = ref_start;
    for (int i = 0; i < n; i++) {
        if (c[i].op >= 9) { *err = AMP_RS_CIGAR_OP; return 0; }
        if (CONSUME_QUERY[c[i].op]) {
            if (query_pos <= cur_pos + c[i].len) {
                if (CONSUME_REF[c[i].op]) ref_pos += query_pos - cur_pos;
                return ref_pos;
            }
            cur_pos += c[i].len;
        }
        if (CONSUME_REF[c[i].op]) ref_pos += c[i].len;
    }
    return ref_pos;
}

/* ---- pysam accessors (unpinned) ---- */
static int64_t reference_length(const cigop *c, int n) {
    int64_t r = 0;
    for (int i = 0; i < n; i++)
        if (c[i].op == OP_M || c[i].op == OP_D || c[i].op == OP_N || c[i].op == OP_EQ || c[i].op == OP_X) r += c[i].len;
    return r ? r : 1; /* htslib bam_endpos */
}
static int64_t query_alignment_start(const cigop *c, int n, int64_t lseq, int *err) {
    int64_t off = 0;
    for (int i = 0; i < n; i++) {
        if (c[i].op == OP_H) { if (off != 0 && off != lseq) { *err = AMP_RS_CLIP; return 0; } }
        else if (c[i].op == OP_S) off += c[i].len;
        else break;
    }
    return off;
}
static int64_t query_alignment_end(const cigop *c, int n, int64_t lseq, int *err) {
    int64_t end = lseq;
    if (end == 0) {
        for (int i = 0; i < n; i++)
            if (c[i].op == OP_M || c[i].op == OP_I || c[i].op == OP_EQ || c[i].op == OP_X || (c[i].op == OP_S && end == 0)) end += c[i].len;
        return end;
    }
    for (int k = n - 1; k >= 1; k--) { /* index 0 is never examined */
        if (c[k].op == OP_H) { if (end != lseq) { *err = AMP_RS_CLIP; return 0; } }
        else if (c[k].op == OP_S) end -= c[k].len;
        else break;
    }
    return end;
}
/* Python slice bounds for seq[a:b] on a length-L sequence */
static void py_slice(int64_t a, int64_t b, int64_t L, int64_t *lo, int64_t *hi) {
    if (a < 0) { a += L; if (a < 0) a = 0; } else if (a > L) a = L;
    if (b < 0) { b += L; if (b < 0) b = 0; } else if (b > L) b = L;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

typedef struct {
    int32_t min_quality, window, do_trim, do_count;
    int32_t ref_len, max_primer_len;
    const int32_t *min_start, *max_end; /* -1 = None */
} orc_params;

/* ---- primer clip from one end: the per-op rules shared by A:467-510 and A:524-555.
 * `src` is walked front to back (the caller reverses for the end clip); start_pos receives
 * the reference advance (only meaningful for the start clip). ---- */
static int primer_clip(const ciglist *src, int64_t del_len, ciglist *dst, int64_t *start_pos_out, int track_pos, int *err) {
    int pos_start = 0;
    int64_t start_pos = 0;
    dst->n = 0;
    for (int i = 0; i < src->n; i++) {
        int32_t cig = src->v[i].op; int64_t n = src->v[i].len;
        if (del_len == 0 && pos_start) { if (cl_push(dst, cig, n)) return -1; continue; }
        if (cig >= 9) { *err = AMP_RS_CIGAR_OP; return 0; }
        if (del_len == 0 && CONSUME_QUERY[cig] && CONSUME_REF[cig]) { pos_start = 1; if (cl_push(dst, cig, n)) return -1; continue; }
        int64_t ref_add = 0;
        if (CONSUME_QUERY[cig]) {
            if (del_len >= n) { if (cl_push(dst, OP_S, n)) return -1; }
            else if (0 < del_len && del_len < n) { if (cl_push(dst, OP_S, del_len)) return -1; }
            else { if (cl_push(dst, OP_S, n)) return -1; continue; }
            ref_add = del_len < n ? del_len : n;
            int64_t tmp = n;
            n = n - del_len > 0 ? n - del_len : 0;
            del_len = del_len - tmp > 0 ? del_len - tmp : 0;
            if (n > 0) { if (cl_push(dst, cig, n)) return -1; }
            int32_t last = dst->v[dst->n - 1].op;
            if (del_len == 0 && CONSUME_QUERY[last] && CONSUME_REF[last]) pos_start = 1;
        } else if (CONSUME_REF[cig]) {
            ref_add += n;
        }
        if (track_pos && CONSUME_REF[cig]) start_pos += ref_add;
    }
    if (start_pos_out) *start_pos_out = start_pos;
    return 0;
}

/* ---- quality clip rewrite shared by A:597-622 and A:658-683 (src walked front to back) ---- */
static int quality_clip(const ciglist *src, int64_t del_len, ciglist *dst, int *err) {
    dst->n = 0;
    for (int i = 0; i < src->n; i++) {
        int32_t cig = src->v[i].op; int64_t n = src->v[i].len;
        if (del_len == 0) { if (cl_push(dst, cig, n)) return -1; continue; }
        if (cig == OP_S || cig == OP_H) { if (cl_push(dst, cig, n)) return -1; continue; }
        if (cig >= 9) { *err = AMP_RS_CIGAR_OP; return 0; }
        if (CONSUME_QUERY[cig]) {
            if (del_len >= n) { if (cl_push(dst, OP_S, n)) return -1; }
            else { if (cl_push(dst, OP_S, del_len)) return -1; }
            int64_t tmp = n;
            n = n - del_len > 0 ? n - del_len : 0;
            del_len = del_len - tmp > 0 ? del_len - tmp : 0;
            if (n > 0) { if (cl_push(dst, cig, n)) return -1; }
        }
    }
    return 0;
}

typedef struct {
    ciglist a, b;        /* current CIGAR and scratch */
    int64_t *pq, *pr;    /* aligned pairs, -1 = None */
    int64_t pcap;
} workspace;

/* ---- A:426-687 trim_read.  cur (ws->a) holds the read's CIGAR and is rewritten in place;
 * *ref_start is updated.  Returns the three flags as AMP_TRIM_* bits; status in *err. ---- */
static int trim_read(const orc_params *p, workspace *ws, int64_t *ref_start, int flag, int64_t tlen, int64_t lseq,
                     const uint8_t *qual, int have_qual, int *err) {
    ciglist *cur = &ws->a, *tmp = &ws->b;
    int flags = 0;
    int is_paired = flag & 1, is_reverse = (flag & 0x10) != 0;
    /* A:450-451 table lookups with the PRE-trim coordinates */
    int64_t rs = *ref_start, re = rs + reference_length(cur->v, cur->n);
    if (rs < 0 || rs >= p->ref_len) { *err = AMP_RS_INDEX_REF; return 0; }
    int32_t left_max_primer_end = p->max_end[rs];
    if (re - 1 >= p->ref_len) { *err = AMP_RS_INDEX_REF; return 0; }
    int32_t right_min_primer_start = p->min_start[re - 1];
    int64_t at = tlen < 0 ? -tlen : tlen;
    int isize_flag = (at - p->max_primer_len) > lseq; /* A:452 */

    if (!(is_paired && isize_flag && is_reverse) && left_max_primer_end >= 0) { /* A:460 */
        flags |= AMP_TRIM_PRIMER_START;
        int64_t del = pos_on_query(cur->v, cur->n, (int64_t)left_max_primer_end + 1, *ref_start, err); /* A:463 */
        if (*err) return 0;
        int64_t start_pos = 0;
        if (primer_clip(cur, del, tmp, &start_pos, 1, err)) return -1;
        if (*err) return 0;
        fix_cigar(tmp);                       /* A:513 */
        { ciglist t = *cur; *cur = *tmp; *tmp = t; }
        *ref_start += start_pos;              /* A:514 */
    }
    if (!(is_paired && isize_flag && !is_reverse) && right_min_primer_start >= 0) { /* A:517 */
        flags |= AMP_TRIM_PRIMER_END;
        int64_t del = lseq - pos_on_query(cur->v, cur->n, right_min_primer_start, *ref_start, err); /* A:520 */
        if (*err) return 0;
        cl_reverse(cur);                      /* A:524 reversed(...) */
        if (primer_clip(cur, del, tmp, NULL, 0, err)) return -1;
        if (*err) return 0;
        cl_reverse(tmp); fix_cigar(tmp);      /* A:558 */
        { ciglist t = *cur; *cur = *tmp; *tmp = t; }
    }
    /* A:561 query_alignment_qualities, after the primer edits */
    int64_t qlen = 0, qs = 0;
    if (lseq == 0) { *err = AMP_RS_NO_QUAL; return 0; }       /* len(None) -> TypeError */
    qs = query_alignment_start(cur->v, cur->n, lseq, err); if (*err) return 0;
    int64_t qe = query_alignment_end(cur->v, cur->n, lseq, err); if (*err) return 0;
    if (!have_qual) { *err = AMP_RS_NO_QUAL; return 0; }
    int64_t lo, hi; py_slice(qs, qe, lseq, &lo, &hi);
    const uint8_t *q = qual + lo; qlen = hi - lo;
    int64_t total = 0, true_end = qlen;
    int64_t window = p->window < true_end ? p->window : true_end;   /* A:563 */
    int64_t mq = p->min_quality;
    if (is_reverse) {                                                /* A:566-625 */
        int64_t i = true_end;
        for (int64_t off = 1; off < window; off++) total += q[i - off];
        while (i > 0) {
            if (window > i) window -= 1; else total += q[i - window];
            if (total < mq * window) break;     /* total/window < min_quality, exact in integers */
            total -= q[i - 1]; i -= 1;
        }
        int64_t del = i;
        int64_t start_pos = pos_on_ref(cur->v, cur->n, del + qs - 1, *ref_start, err);  /* A:591 */
        if (*err) return 0;
        if (start_pos > *ref_start) {                                /* A:594 */
            flags |= AMP_TRIM_QUALITY;
            if (quality_clip(cur, del, tmp, err)) return -1;
            if (*err) return 0;
            fix_cigar(tmp);                                          /* A:625; reference_start NOT advanced */
            { ciglist t = *cur; *cur = *tmp; *tmp = t; }
        }
    } else {                                                         /* A:628-686 */
        int64_t i = 0;
        for (int64_t off = 0; off < window - 1; off++) total += q[i + off];
        while (i < true_end) {
            if (true_end - window < i) window -= 1; else total += q[i + window - 1];
            if (total < mq * window) break;
            total -= q[i]; i += 1;
        }
        int64_t del = true_end - i;
        (void)pos_on_ref(cur->v, cur->n, del, *ref_start, err);      /* A:653: value unused, may raise */
        if (*err) return 0;
        if (del != 0) {
            flags |= AMP_TRIM_QUALITY;
            cl_reverse(cur);
            if (quality_clip(cur, del, tmp, err)) return -1;
            if (*err) return 0;
            cl_reverse(tmp); fix_cigar(tmp);                         /* A:686 */
            { ciglist t = *cur; *cur = *tmp; *tmp = t; }
        }
    }
    return flags;
}

typedef struct {
    amp_ins_event *v;
    int64_t n, cap;
} evlist;

static int ev_push(evlist *e, int32_t pos, uint32_t read, int64_t from, int64_t to) {
    if (e->n == e->cap) {
        int64_t nc = e->cap ? 2 * e->cap : 1024;
        amp_ins_event *nv = (amp_ins_event *)realloc(e->v, (size_t)nc * sizeof(*nv));
        if (!nv) return -1;
        e->v = nv; e->cap = nc;
    }
    e->v[e->n].ref_pos = pos; e->v[e->n].read = read; e->v[e->n].q_from = (int32_t)from; e->v[e->n].q_to = (int32_t)to;
    e->n++;
    return 0;
}

/* BAM 4-bit code -> count-table column, -1 = KeyError (A:892 has keys A C G T N '-') */
static const int8_t CODE2COL[16] = {-1, 0, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, 4};

static inline int base_code(const uint8_t *seq, int64_t base_off, int64_t q) {
    int64_t k = base_off + q;
    uint8_t b = seq[k >> 1];
    return (k & 1) ? (b & 15) : (b >> 4);
}

/* ---- A:690-753 update_base_counts on the (possibly trimmed) CIGAR in ws->a ---- */
static int update_base_counts(const orc_params *p, workspace *ws, int64_t ref_start, int64_t lseq, const uint8_t *seq,
                              int64_t base_off, const uint8_t *qual, int have_qual, uint32_t *counts, evlist *ev,
                              uint32_t read_id, int *err) {
    const ciglist *c = &ws->a;
    int64_t query_start = query_alignment_start(c->v, c->n, lseq, err); if (*err) return 0;   /* A:700 */
    int64_t query_end = query_alignment_end(c->v, c->n, lseq, err); if (*err) return 0;       /* A:701 */
    if (lseq == 0) { *err = AMP_RS_NO_SEQ; return 0; }                                         /* A:702 */
    int64_t ref_end = ref_start + reference_length(c->v, c->n);                                /* A:705 */
    /* A:706 get_aligned_pairs */
    int64_t np = 0;
    for (int i = 0; i < c->n; i++) if (c->v[i].op != OP_H && c->v[i].op < 9) np += c->v[i].len;
    if (np > ws->pcap) {
        int64_t nc = np + 64;
        int64_t *a = (int64_t *)realloc(ws->pq, (size_t)nc * sizeof(int64_t)); if (!a) return -1; ws->pq = a;
        int64_t *b = (int64_t *)realloc(ws->pr, (size_t)nc * sizeof(int64_t)); if (!b) return -1; ws->pr = b;
        ws->pcap = nc;
    }
    int64_t *pq = ws->pq, *pr = ws->pr;
    {
        int64_t k = 0, q = 0, r = ref_start;
        for (int i = 0; i < c->n; i++) {
            int32_t op = c->v[i].op; int64_t n = c->v[i].len;
            if (op == OP_M || op == OP_EQ || op == OP_X) { for (int64_t j = 0; j < n; j++) { pq[k] = q++; pr[k++] = r++; } }
            else if (op == OP_I || op == OP_S || op == OP_P) { for (int64_t j = 0; j < n; j++) { pq[k] = q++; pr[k++] = -1; } }
            else if (op == OP_D || op == OP_N) { for (int64_t j = 0; j < n; j++) { pq[k] = -1; pr[k++] = r++; } }
        }
        np = k;
    }
    int64_t mq = p->min_quality, G = p->ref_len;
    int64_t i = 0;
    while (i < np) {
        int64_t q_pos = pq[i], r_pos = pr[i]; i++;
        if (q_pos < 0) {                                             /* A:714-715 deletion / ref skip */
            if (r_pos >= G) { *err = AMP_RS_INDEX_REF; return 0; }
            counts[r_pos * AMP_NSYM + 5] += 1;
            continue;
        }
        if (!have_qual) { *err = AMP_RS_NO_QUAL; return 0; }         /* None[q] */
        if (q_pos >= lseq) { *err = AMP_RS_INDEX_QUERY; return 0; }
        if (qual[q_pos] < mq) continue;                              /* A:718 */
        if (q_pos < query_start) continue;                           /* A:722 */
        if (q_pos >= query_end) break;                               /* A:726 */
        if (r_pos < 0) {                                             /* A:730 insertion */
            int64_t q0 = q_pos;
            int q_none = 0;
            while (r_pos < 0 && !q_none && q_pos < query_end) {
                if (q_pos >= lseq) { *err = AMP_RS_INDEX_QUERY; return 0; }
                if (!(qual[q_pos] >= mq)) break;
                if (i >= np) { *err = AMP_RS_INDEX_PAIRS; return 0; } /* A:734 */
                q_pos = pq[i]; r_pos = pr[i]; i++;
                if (q_pos < 0) q_none = 1;
            }
            /* (when q_none, r_pos is a real coordinate, so the Python loop also stops on `r_pos is None`) */
            int64_t lo, hi;
            if (r_pos == 0) {                                        /* A:735-736 */
                if (q_none) { *err = AMP_RS_TYPE; return 0; }        /* None + 1 */
                py_slice(q0, q_pos + 1, lseq, &lo, &hi);
            } else {
                if (q_none) py_slice(q0 - 1, lseq, lseq, &lo, &hi);  /* seq[a:None] */
                else py_slice(q0 - 1, q_pos, lseq, &lo, &hi);        /* A:738 */
            }
            int64_t ins_pos;
            if (r_pos < 0) ins_pos = ref_end;                        /* A:739-740 */
            else { ins_pos = r_pos; i -= 1; }                        /* A:742-743 */
            ins_pos = ins_pos - 1 > 0 ? ins_pos - 1 : 0;             /* A:744 */
            if (ins_pos >= G) { *err = AMP_RS_INDEX_REF; return 0; }
            if (ev_push(ev, (int32_t)ins_pos, read_id, lo, hi)) return -1;
            continue;
        }
        /* A:751-753 match / mismatch */
        if (r_pos >= G) { *err = AMP_RS_INDEX_REF; return 0; }
        int col = CODE2COL[base_code(seq, base_off, q_pos)];
        if (col < 0) { *err = AMP_RS_KEY_BASE; return 0; }
        counts[r_pos * AMP_NSYM + col] += 1;
    }
    (void)ref_end;
    return 0;
}

/* =====================================================================================
 * exported entry points (loaded with ctypes by the tests and by bench.py)
 * ===================================================================================== */

/* A:174-209 find_overlapping_primers: the deque sweep, literally. */
int orc_find_overlapping_primers(int32_t ref_len, int32_t n, const int32_t *starts, const int32_t *ends, int32_t off,
                                 int32_t *min_start, int32_t *max_end, int32_t *max_primer_len) {
    int32_t *dq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (!dq) return AMP_ENOMEM;
    int head = 0, tail = 0, i = 0;
    for (int32_t p = 0; p < ref_len; p++) {
        while (head != tail && p >= ends[dq[head]] + off) head++;
        while (i < n && p >= starts[i] - off) dq[tail++] = i++;
        if (head != tail) {
            int32_t mn = starts[dq[head]], mx = ends[dq[head]];
            for (int k = head + 1; k < tail; k++) { if (starts[dq[k]] < mn) mn = starts[dq[k]]; if (ends[dq[k]] > mx) mx = ends[dq[k]]; }
            min_start[p] = mn; max_end[p] = mx;
        } else { min_start[p] = -1; max_end[p] = -1; }
    }
    int32_t mpl = 0;
    for (int k = 0; k < n; k++) if (k == 0 || ends[k] - starts[k] > mpl) mpl = ends[k] - starts[k];  /* A:876 */
    if (max_primer_len) *max_primer_len = mpl;
    free(dq);
    return AMP_OK;
}

int64_t orc_pos_on_query(int32_t n, const int32_t *ops, const int64_t *lens, int64_t ref_pos, int64_t ref_start, int *err) {
    cigop *c = (cigop *)malloc(sizeof(cigop) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; i++) { c[i].op = ops[i]; c[i].len = lens[i]; }
    *err = 0;
    int64_t r = pos_on_query(c, n, ref_pos, ref_start, err);
    free(c);
    return r;
}
int64_t orc_pos_on_ref(int32_t n, const int32_t *ops, const int64_t *lens, int64_t query_pos, int64_t ref_start, int *err) {
    cigop *c = (cigop *)malloc(sizeof(cigop) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; i++) { c[i].op = ops[i]; c[i].len = lens[i]; }
    *err = 0;
    int64_t r = pos_on_ref(c, n, query_pos, ref_start, err);
    free(c);
    return r;
}
int32_t orc_fix_cigar(int32_t n, int32_t *ops, int64_t *lens) {
    ciglist c = {0};
    for (int i = 0; i < n; i++) cl_push(&c, ops[i], lens[i]);
    fix_cigar(&c);
    for (int i = 0; i < c.n; i++) { ops[i] = c.v[i].op; lens[i] = c.v[i].len; }
    int32_t r = c.n;
    free(c.v);
    return r;
}

/* A:896-915 for rows [lo, hi) of a packed batch.  counts is uint32[ref_len][6], added to.
 * Events are appended to a malloc'ed list returned through ev_out / n_ev (free with orc_free). */
int orc_process_range(int32_t min_quality, int32_t window, int32_t do_trim, int32_t do_count, int32_t ref_len,
                      const int32_t *min_start, const int32_t *max_end, int32_t max_primer_len,
                      const amp_reads *rd, int64_t lo, int64_t hi, uint64_t read_base, const amp_trim_out *out,
                      uint32_t *counts, amp_ins_event **ev_out, int64_t *n_ev) {
    orc_params p = {min_quality, window, do_trim, do_count, ref_len, max_primer_len, min_start, max_end};
    if (window < 1 || min_quality < 0 || ref_len < 0) return AMP_EINVAL;
    workspace ws; memset(&ws, 0, sizeof(ws));
    evlist ev = {0};
    int rc = AMP_OK;
    for (int64_t i = lo; i < hi; i++) {
        int64_t c0 = (int64_t)rd->cig_off[i], c1 = (int64_t)rd->cig_off[i + 1];
        int ncig = (int)(c1 - c0);
        ws.a.n = 0;
        if (cl_reserve(&ws.a, ncig + 4) || cl_reserve(&ws.b, ncig + 4)) { rc = AMP_ENOMEM; break; }
        for (int k = 0; k < ncig; k++) { ws.a.v[k].op = (int32_t)(rd->cig[c0 + k] & 15u); ws.a.v[k].len = (int64_t)(rd->cig[c0 + k] >> 4); }
        ws.a.n = ncig;
        int64_t ref_start = rd->pos[i];
        int64_t lseq = rd->lseq[i];
        int64_t boff = (int64_t)rd->seq_off[i];
        const uint8_t *qual = rd->qual + boff;
        int have_qual = lseq > 0 && qual[0] != 0xFF;
        int err = 0, flags = 0;
        if (do_trim) {
            flags = trim_read(&p, &ws, &ref_start, rd->flag[i], rd->tlen[i], lseq, qual, have_qual, &err);
            if (flags < 0) { rc = AMP_ENOMEM; break; }
        }
        if (out) {
            if (out->new_pos) out->new_pos[i] = (int32_t)ref_start;
            if (out->new_ncig) out->new_ncig[i] = err ? 0 : (uint32_t)ws.a.n;
            if (out->new_cig && !err) for (int k = 0; k < ws.a.n; k++) out->new_cig[c0 + 3 * i + k] = ((uint32_t)ws.a.v[k].len << 4) | (uint32_t)ws.a.v[k].op;
            if (out->ref_len) out->ref_len[i] = err ? 0 : (int32_t)reference_length(ws.a.v, ws.a.n);
            if (out->trim_flags) out->trim_flags[i] = err ? 0 : (uint8_t)flags;
        }
        if (!err && do_count) {
            int r = update_base_counts(&p, &ws, ref_start, lseq, rd->seq, boff, qual, have_qual, counts, &ev,
                                       (uint32_t)(read_base + (uint64_t)i), &err);
            if (r < 0) { rc = AMP_ENOMEM; break; }
        }
        if (out && out->status) out->status[i] = (uint8_t)err;
    }
    free(ws.a.v); free(ws.b.v); free(ws.pq); free(ws.pr);
    if (ev_out) { *ev_out = ev.v; *n_ev = ev.n; } else free(ev.v);
    return rc;
}

void orc_free(void *p) { free(p); }

/* ---- A:756-771 alleles_from_counts + A:917-952, for ONE position.
 * syms: n symbol strings (the 6 base symbols plus this position's insertion strings) with
 * counts.  Writes the indices of the non-zero symbols ranked like
 * sorted(((count, count/total, k) ...), reverse=True): count desc, then string desc
 * (Python compares code points; ASCII here).  Returns the number of ranked alleles. ---- */
static int py_str_cmp(const char *a, const char *b) {
    const unsigned char *x = (const unsigned char *)a, *y = (const unsigned char *)b;
    while (*x && *x == *y) { x++; y++; }
    return (int)*x - (int)*y;
}
int32_t orc_rank_alleles(int32_t n, const char **syms, const uint32_t *cnt, int32_t *order, uint64_t *total_out) {
    uint64_t total = 0;
    int32_t m = 0;
    for (int i = 0; i < n; i++) { total += cnt[i]; if (cnt[i]) order[m++] = i; }
    for (int i = 1; i < m; i++) { /* insertion sort, descending */
        int32_t v = order[i]; int j = i - 1;
        while (j >= 0 && (cnt[order[j]] < cnt[v] || (cnt[order[j]] == cnt[v] && py_str_cmp(syms[order[j]], syms[v]) < 0))) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    *total_out = total;
    return m;
}
