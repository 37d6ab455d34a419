// amp_bf.hpp -- the closed-form trims of amp_read.hpp (Cig2: reads of the shape [S a][M m1]([I|D k][M m2])[S c]) once
// more, written WITHOUT branches: every rule of primer_clip / quality_clip (A:467-510, A:524-555, A:597-622, A:658-683),
// get_pos_on_query (A:389-412) and get_pos_on_ref (A:363-386) becomes arithmetic and selects on the five lengths.
//
// Why: one lane per read means a wave executes the union of all the paths its 64 reads take, and every `if' of the
// branchy forms costs a save / branch / restore of the execution mask even when it is not taken; the per-read logic was
// a third of the fast kernel's instructions.  Here a read's whole trim is ~300 straight-line vector instructions.
// tests/hostsim fuzzes these forms against the branchy ones and against the generic code on the CPU
// (tests/test_hostsim_golden.py::test_branch_free_closed_forms).
#pragma once

#include "amp_read.hpp"

namespace amp {

struct Bf {                       // the shape; kind 0 none (k = m2 = 0), 1 insertion, 2 deletion
    int32_t a, m1, k, m2, c, kind;
    uint32_t op;
    uint32_t punt;                // 1: a shape the closed forms leave to the generic code
    AMP_HD int32_t kI() const { return kind == 1 ? k : 0; }
    AMP_HD int32_t kD() const { return kind == 2 ? k : 0; }
    AMP_HD int32_t query_len() const { return a + m1 + kI() + m2 + c; }
    AMP_HD int32_t ref_len() const { const int32_t r = m1 + m2 + kD(); return r ? r : 1; }
};
AMP_HD int32_t bf_sel(bool c, int32_t x, int32_t y) { return c ? x : y; }
AMP_HD int32_t bf_min(int32_t x, int32_t y) { return x < y ? x : y; }
AMP_HD int32_t bf_max(int32_t x, int32_t y) { return x > y ? x : y; }
// c ? x : y, field by field (a select of whole structs goes through memory)
AMP_HD Bf bf_pick(bool c, const Bf &x, const Bf &y) {
    return Bf{c ? x.a : y.a, c ? x.m1 : y.m1, c ? x.k : y.k, c ? x.m2 : y.m2, c ? x.c : y.c, c ? x.kind : y.kind, c ? x.op : y.op, c ? x.punt : y.punt};
}
AMP_HD Bf bf_mirror(const Bf &s) { return Bf{s.c, s.kind ? s.m2 : s.m1, s.k, s.kind ? s.m1 : s.m2, s.a, s.kind, s.op, s.punt}; }

// the shape of an input CIGAR of n ops (its first five words; words past the read's own may hold anything) -- cig2_from_words5;
// ok = 0: not a shape of the family
AMP_HD Bf bf_from_words5(int n, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, int32_t lseq, int32_t max_ins, int32_t max_del, bool &ok) {
    const bool lead = (w0 & 15u) == OP_S;
    const uint32_t v0 = lead ? w1 : w0, v1 = lead ? w2 : w1, v2 = lead ? w3 : w2, v3 = lead ? w4 : w3;
    const int m = n - (lead ? 1 : 0);                      // ops behind the leading clip: M | M S | M X M | M X M S
    const uint32_t o0 = v0 & 15u, o1 = v1 & 15u;
    const bool has3 = m >= 3, isI = o1 == OP_I, isD = o1 == OP_D;
    const bool has_tail = has3 ? m == 4 : m == 2;
    const uint32_t tail = has3 ? v3 : v1;
    Bf s;
    s.op = o0; s.punt = 0u;
    s.a = lead ? (int32_t)(w0 >> 4) : 0;
    s.m1 = (int32_t)(v0 >> 4);
    s.kind = has3 ? (isI ? 1 : 2) : 0;
    s.k = has3 ? (int32_t)(v1 >> 4) : 0;
    s.m2 = has3 ? (int32_t)(v2 >> 4) : 0;
    s.c = has_tail ? (int32_t)(tail >> 4) : 0;
    bool good = (n >= 1) & (n <= 5) & (lseq > 0) & (m >= 1) & (m <= 4) & ((o0 == OP_M) | (o0 == OP_EQ) | (o0 == OP_X)) & (s.m1 > 0);
    good = good & (!lead | (s.a > 0));
    good = good & (!has3 | ((isI | isD) & ((v2 & 15u) == o0) & (s.k > 0) & (s.m2 > 0)));
    good = good & (!has_tail | (((tail & 15u) == OP_S) & (s.c > 0)));
    good = good & (s.query_len() == lseq);
    good = good & ((s.kind != 1) | (s.k <= max_ins)) & ((s.kind != 2) | (s.k <= max_del));
    ok = good;
    return bf_pick(good, s, Bf{0, 0, 0, 0, 0, 0, 0u, 0u});
}

// get_pos_on_query (A:389-412) on the shape; t = ref_pos - reference_start
AMP_HD int32_t bf_pos_on_query(const Bf &s, int32_t t) {
    const bool in1 = (s.m1 > 0) & (t <= s.m1);
    const int32_t tD = t - s.m1;
    const bool inD = (s.kind == 2) & (tD <= s.k) & !in1;
    const int32_t u = tD - s.kD();
    const bool in2 = (s.kind != 0) & (s.m2 > 0) & (u <= s.m2) & !in1 & !inD;
    int32_t r = s.query_len();
    r = in2 ? s.a + s.m1 + s.kI() + u : r;
    r = inD ? s.a + s.m1 : r;
    r = in1 ? s.a + t : r;
    return r;
}

// get_pos_on_ref (A:363-386) on the shape, minus reference_start
AMP_HD int32_t bf_ref_offset(const Bf &s, int32_t qp) {
    const bool in0 = (s.a > 0) & (qp <= s.a);
    const int32_t x = qp - s.a;
    const bool in1 = !in0 & (s.m1 > 0) & (x <= s.m1);
    const int32_t x2 = x - s.m1;
    const bool inI = !in0 & !in1 & (s.kind == 1) & (x2 <= s.k);
    const int32_t x3 = x2 - s.kI();
    const bool in2 = !in0 & !in1 & !inI & (s.kind != 0) & (s.m2 > 0) & (x3 <= s.m2);
    int32_t r = s.m1 + s.kD() + (s.kind ? s.m2 : 0);
    r = in2 ? s.m1 + s.kD() + x3 : r;
    r = inI ? s.m1 : r;
    r = in1 ? x : r;
    r = in0 ? 0 : r;
    return r;
}

// d >= 0 query bases behind the soft clip become soft clip (primer_clip with its `del' already reduced by the clip's share,
// A:467-510, or quality_clip, A:597-622: QUALITY).  What the two differ in: a clip that ends exactly in front of the indel
// keeps it when it is a quality clip (the op is copied behind the clip) and swallows it when it is a primer clip (an
// insertion turns into soft clip, a deletion is dropped and the start jumps over it).  adv = the reference advance.
template <bool QUALITY>
AMP_HD Bf bf_clip_left(const Bf &s, int32_t d, int32_t &adv) {
    const bool r1 = (s.m1 > 0) & (d < s.m1);                                  // the clip ends inside the first match op
    const int32_t d1 = d - s.m1, A = s.a + s.m1;
    const bool stopI = !r1 & (s.kind == 1) & (d1 < s.k) & (QUALITY | (d1 > 0));     // ... inside the insertion (or right in front of it)
    const bool stopD = !r1 & (s.kind == 2) & (d1 == 0) & QUALITY;                  // ... right in front of the deletion
    const bool stop = stopI | stopD;
    // the clip goes through the indel: what is left of the second segment
    const int32_t d2 = s.kind == 1 ? bf_max(d1 - s.k, 0) : d1;
    const int32_t m = s.kind ? s.m2 : 0;
    const int32_t t = bf_min(d2, m);
    const int32_t m_left = m - t;
    const int32_t A4 = A + s.kI() + t;
    Bf o;
    o.op = s.op;
    o.punt = s.punt | ((!r1 & (s.kind != 0) & (s.m2 == 0)) ? 1u : 0u);          // an indel that already touches the far clip
    o.a = r1 ? s.a + d : (stop ? A + (stopI ? d1 : 0) : A4 + (m_left == 0 ? s.c : 0));
    o.m1 = r1 ? s.m1 - d : (stop ? 0 : m_left);
    o.k = r1 ? s.k : (stop ? s.k - (stopI ? d1 : 0) : 0);
    o.m2 = (r1 | stop) ? s.m2 : 0;
    o.kind = (r1 | stop) ? s.kind : 0;
    o.c = (r1 | stop) ? s.c : (m_left == 0 ? 0 : s.c);
    adv = r1 ? d : (stop ? s.m1 : s.m1 + s.kD() + t);
    return o;
}

// primer_clip (A:467-510) of `del' query bases from the front; returns the reference advance
AMP_HD Bf bf_primer_clip(const Bf &s, int32_t del, int32_t &adv) {
    const int32_t d = s.a > 0 ? bf_max(del - s.a, 0) : del;                    // a soft clip stays one and eats its share
    int32_t adv_c;
    const Bf c = bf_clip_left<false>(s, bf_max(d, 0), adv_c);
    const bool neg = del < 0, zero = del == 0;
    // del < 0: every query-consuming op falls into the "else" arm -- all soft clip; a deletion still advances
    Bf o;
    o.op = s.op;
    o.punt = (neg | zero) ? s.punt : c.punt;
    o.a = neg ? s.query_len() : (zero ? s.a : c.a);
    o.m1 = neg ? 0 : (zero ? s.m1 : c.m1);
    o.k = neg ? 0 : (zero ? s.k : c.k);
    o.m2 = neg ? 0 : (zero ? s.m2 : c.m2);
    o.c = neg ? 0 : (zero ? s.c : c.c);
    o.kind = neg ? 0 : (zero ? s.kind : c.kind);
    adv = neg ? s.kD() : (zero ? 0 : adv_c);
    return o;
}
AMP_HD Bf bf_canon(const Bf &s) {         // no match left: one soft clip
    const bool all = (s.kind == 0) & (s.m1 == 0);
    Bf o = s;
    o.a = all ? s.a + s.c : s.a;
    o.c = all ? 0 : s.c;
    return o;
}

// Stage 1+2 of trim_read (A:450-558) on the shape, given the two table entries (-1 = None) and the outcome of the
// template-length test of A:452: cig2_trim_primers_isize
AMP_HD Bf bf_trim_primers(const Bf &s0, int32_t &pos, uint32_t &flags, uint32_t flag, bool isize_flag, int32_t lseq,
                          int32_t left_max_end, int32_t right_min_start) {
    const bool is_paired = flag & 1u, is_reverse = (flag & 0x10u) != 0;
    const bool do1 = !(is_paired & isize_flag & is_reverse) & (left_max_end >= 0);          // A:460
    int32_t adv1;
    const Bf c1 = bf_primer_clip(s0, bf_pos_on_query(s0, left_max_end + 1 - pos), adv1);   // A:463
    Bf s = bf_pick(do1, c1, s0);
    pos += do1 ? adv1 : 0;                                                                 // A:514
    flags |= do1 ? AMP_TRIM_PRIMER_START : 0u;
    const bool do2 = !(is_paired & isize_flag & !is_reverse) & (right_min_start >= 0) & !s.punt;   // A:517
    const int32_t del2 = lseq - bf_pos_on_query(s, right_min_start - pos);                 // A:520
    int32_t adv2;
    const Bf c2 = bf_canon(bf_mirror(bf_primer_clip(bf_mirror(s), del2, adv2)));
    flags |= do2 ? AMP_TRIM_PRIMER_END : 0u;
    return bf_pick(do2, c2, s);
}

// aligned-quality window of the shape: [lo, lo + qlen) -- cig2_quality_window
AMP_HD void bf_quality_window(const Bf &s, int32_t lseq, int32_t &lo, int32_t &qlen) {
    const bool single = (s.m1 == 0) & (s.kind == 0);        // one soft clip covering the read: its end is not examined
    const int32_t qe = single ? lseq : lseq - s.c;
    lo = bf_min(s.a, lseq);
    qlen = bf_max(qe, lo) - lo;
}

// Stage 3 of trim_read given the scan result i (A:589-625, A:651-686) -- cig2_trim_quality
AMP_HD Bf bf_trim_quality(const Bf &s, int32_t pos, uint32_t &flags, bool is_reverse, int32_t i, int32_t qlen) {
    int32_t adv;
    // reverse strand: trimmed only if get_pos_on_ref(del + query_alignment_start - 1) > reference_start (A:591-594);
    // reference_start is NOT advanced
    const int32_t del_r = i;
    const bool do_r = is_reverse & (bf_ref_offset(s, del_r + s.a - 1) > 0);
    const Bf cr = bf_clip_left<true>(s, del_r, adv);
    const Bf cr0 = bf_pick(del_r == 0, s, cr);                     // (quality_clip returns at once on del == 0)
    const int32_t del_f = qlen - i;
    const bool do_f = !is_reverse & (del_f != 0);                                          // A:656
    const Bf cf = bf_canon(bf_mirror(bf_clip_left<true>(bf_mirror(s), del_f, adv)));
    flags |= (do_r | do_f) ? AMP_TRIM_QUALITY : 0u;
    (void)pos;
    return bf_pick(do_r, cr0, bf_pick(do_f, cf, s));
}

}  // namespace amp
