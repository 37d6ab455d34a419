// hostsim -- TEST INFRASTRUCTURE.  Runs the per-read device functions of
// amplipy_amd/csrc/amp_read.hpp on the CPU so that their logic can be checked against the
// golden vectors in a container without a GPU.  Never loaded by the amplipy_amd package
// and not a fallback: the product library has no host execution path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../amplipy_amd/csrc/amp_read.hpp"
#include "../../amplipy_amd/csrc/amp_bf.hpp"

using namespace amp;

struct HostSink {
    uint32_t *counts;
    std::vector<amp_ins_event> *ev;
    uint32_t read;
    void add(int32_t r, uint32_t col) { counts[(size_t)r * AMP_NSYM + col] += 1; }
    void event(int32_t pos, int32_t lo, int32_t hi) { ev->push_back(amp_ins_event{pos, read, lo, hi}); }
};

extern "C" int sim_process_range(int32_t min_quality, int32_t window, int32_t do_trim, int32_t do_count, int32_t ref_len,
                                 const int32_t *min_start, const int32_t *max_end, int32_t max_primer_len,
                                 const amp_reads *rd, int64_t lo, int64_t hi, uint64_t read_base, const amp_trim_out *out,
                                 uint32_t *counts, amp_ins_event **ev_out, int64_t *n_ev) {
    KParams P{min_quality, window, do_trim, do_count, ref_len, max_primer_len, min_start, max_end};
    std::vector<amp_ins_event> ev;
    std::vector<uint32_t> a, b;
    for (int64_t i = lo; i < hi; ++i) {
        int64_t c0 = (int64_t)rd->cig_off[i], c1 = (int64_t)rd->cig_off[i + 1];
        int n = (int)(c1 - c0);
        a.assign(n + 3, 0); b.assign(n + 3, 0);
        memcpy(a.data(), rd->cig + c0, sizeof(uint32_t) * n);
        CigBuf<1> cur{a.data()}, tmp{b.data()};
        int32_t lseq = (int32_t)rd->lseq[i];
        int64_t boff = (int64_t)rd->seq_off[i];
        const uint8_t *qual = rd->qual + boff;
        bool have_qual = lseq > 0 && qual[0] != 0xFF;
        TrimState st{rd->pos[i], n, 0u, 0};
        if (do_trim) trim_read_serial(P, st, rd->flag[i], rd->tlen[i], lseq, qual, have_qual, cur, tmp);
        if (out) {
            if (out->new_pos) out->new_pos[i] = st.pos;
            if (out->new_ncig) out->new_ncig[i] = st.err ? 0 : st.n;
            if (out->new_cig && !st.err) for (int k = 0; k < st.n; ++k) out->new_cig[c0 + 3 * i + k] = cur.get(k);
            if (out->ref_len) out->ref_len[i] = st.err ? 0 : reference_length(cur, st.n);
            if (out->trim_flags) out->trim_flags[i] = st.err ? 0 : (uint8_t)st.flags;
        }
        int err = st.err;
        if (!err && do_count) {
            HostSink sink{counts, &ev, (uint32_t)(read_base + (uint64_t)i)};
            err = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytesCached{rd->seq, boff, qual}, have_qual, sink);
        }
        if (out && out->status) out->status[i] = (uint8_t)err;
    }
    *n_ev = (int64_t)ev.size();
    *ev_out = nullptr;
    if (!ev.empty()) {
        *ev_out = (amp_ins_event *)malloc(sizeof(amp_ins_event) * ev.size());
        memcpy(*ev_out, ev.data(), sizeof(amp_ins_event) * ev.size());
    }
    return 0;
}
extern "C" void sim_free(void *p) { free(p); }


// ---- fuzz of the two-segment closed forms (Cig2) against the generic trim code ------------------------------
static uint64_t rng_state;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }
static int32_t rin(int32_t lo, int32_t hi) { return lo + (int32_t)(rnd() % (uint32_t)(hi - lo + 1)); }

// returns the number of mismatches; *n_punt / *n_cmp report how many cases were punted / compared
extern "C" long sim_cig2_fuzz(uint64_t seed, long iters, long *n_punt, long *n_cmp) {
    rng_state = seed * 0x9E3779B97F4A7C15ull + 12345;
    const int32_t G = 2000;
    std::vector<int32_t> mn(G), mx(G);
    long bad = 0; *n_punt = 0; *n_cmp = 0;
    for (long it = 0; it < iters; ++it) {
        // a few primers, tables like find_overlapping_primers builds them (offset 0..3)
        const int32_t off = rin(0, 3);
        for (int p = 0; p < G; ++p) { mn[p] = -1; mx[p] = -1; }
        int32_t mpl = 0;
        const int npr = rin(1, 4);
        for (int j = 0; j < npr; ++j) {
            const int32_t a = rin(0, 400), b = a + rin(1, 40);
            mpl = b - a > mpl ? b - a : mpl;
            for (int32_t p = a - off; p < b + off; ++p)
                if (p >= 0 && p < G) { mn[p] = mn[p] < 0 || a < mn[p] ? a : mn[p]; mx[p] = b > mx[p] ? b : mx[p]; }
        }
        const int32_t window = rin(1, 8), mq = rin(0, 40);
        KParams P{mq, window, 1, 1, G, mpl, mn.data(), mx.data()};
        // the read
        const int kind = rin(0, 2);
        const int32_t m1 = rin(1, 60), k = kind ? rin(1, 6) : 0, m2 = kind ? rin(1, 60) : 0;
        const uint32_t op = (rnd() % 8 == 0) ? OP_EQ : OP_M;
        // (a third of the reads arrive with soft clips at one or both ends, as aligners leave them)
        const int32_t sa = rnd() % 3 == 0 ? rin(1, 30) : 0, sc = rnd() % 3 == 0 ? rin(1, 30) : 0;
        uint32_t in[5] = {0, 0, 0, 0, 0}; int n = 0;
        if (sa) in[n++] = ((uint32_t)sa << 4) | OP_S;
        in[n++] = ((uint32_t)m1 << 4) | op;
        if (kind) { in[n++] = ((uint32_t)k << 4) | (kind == 1 ? OP_I : OP_D); in[n++] = ((uint32_t)m2 << 4) | op; }
        if (sc) in[n++] = ((uint32_t)sc << 4) | OP_S;
        const int32_t lseq = sa + m1 + (kind == 1 ? k : 0) + m2 + sc;
        const int32_t pos = rin(0, 420);
        const uint32_t flag = (rnd() & 1u) | ((rnd() & 1u) << 4);
        const int32_t tlen = (rnd() & 1) ? rin(-600, 600) : 0;
        std::vector<uint8_t> qual(lseq + 16);
        const int mode = rin(0, 3);
        for (int32_t q = 0; q < lseq; ++q) {
            uint8_t v = (uint8_t)rin(mode == 0 ? 30 : 0, 41);
            if (mode == 2 && q > lseq - rin(1, 20)) v = 2;
            if (mode == 3 && q < rin(0, 20)) v = 2;
            qual[q] = v;
        }
        // generic
        uint32_t a[8] = {0}, b[8] = {0};
        memcpy(a, in, sizeof(uint32_t) * n);
        CigBuf<1> cur{a}, tmp{b};
        TrimState st{pos, n, 0u, 0};
        trim_read_serial(P, st, flag, tlen, lseq, qual.data(), true, cur, tmp);
        // closed forms
        Cig2 s;
        if (!cig2_from_words5(n, in, lseq, s)) { ++bad; continue; }
        if (!sa && !sc) {       // the three-word form agrees on inputs without clips
            Cig2 s3;
            if (!cig2_from_words(n, in[0], n > 1 ? in[1] : 0u, n > 2 ? in[2] : 0u, lseq, s3) || s3.m1 != s.m1 || s3.k != s.k || s3.m2 != s.m2 || s3.kind != s.kind) { ++bad; continue; }
        }
        TrimState t2{pos, n, 0u, 0};
        const int32_t rs = pos, re1 = pos + s.ref_len() - 1;
        if ((uint32_t)rs >= (uint32_t)G || (uint32_t)re1 >= (uint32_t)G) t2.err = AMP_RS_INDEX_REF;
        else {
            cig2_trim_primers(P, t2, flag, tlen, lseq, s, mx[rs], mn[re1]);
            if (!s.punt) {
                int32_t lo, qlen;
                cig2_quality_window(s, lseq, lo, qlen);
                const int32_t iq = quality_scan(qual.data() + lo, qlen, window, mq, (flag & 0x10u) != 0);
                cig2_trim_quality(t2, (flag & 0x10u) != 0, iq, qlen, s);
            }
        }
        if (s.punt) { ++*n_punt; continue; }
        ++*n_cmp;
        uint32_t c2[8]; CigBuf<1> cb{c2};
        const int n2 = t2.err ? 0 : s.store(cb);
        bool same = st.err == t2.err;
        if (same && !st.err) {
            same = st.pos == t2.pos && st.flags == t2.flags && st.n == n2;
            for (int j = 0; same && j < n2; ++j) same = cur.get(j) == c2[j];
        }
        if (same && !st.err) {
            // counting: generic exact walk against segments + cig2_indels
            std::vector<uint8_t> seq((lseq + 1) / 2 + 8);
            for (auto &x : seq) { const uint8_t c[5] = {1, 2, 4, 8, 15}; x = (uint8_t)((c[rnd() % (rnd() % 16 ? 4 : 5)] << 4) | c[rnd() % 4]); }
            std::vector<uint32_t> c1((size_t)G * AMP_NSYM, 0), cc2((size_t)G * AMP_NSYM, 0);
            std::vector<amp_ins_event> e1, e2;
            HostSink s1{c1.data(), &e1, 7u}, s2{cc2.data(), &e2, 7u};
            const int er1 = count_read_walk(P, cur, st.n, st.pos, lseq, ReadBytes{seq.data(), 0, qual.data()}, true, s1);
            int er2 = 0;
            {
                ReadBytes rb{seq.data(), 0, qual.data()};
                const int32_t qa[2] = {s.a, s.a + s.m1 + (s.kind == 1 ? s.k : 0)}, ml[2] = {s.m1, s.kind ? s.m2 : 0};
                const int32_t r0[2] = {t2.pos, t2.pos + s.m1 + (s.kind == 2 ? s.k : 0)};
                for (int sg = 0; sg < 2 && !er2; ++sg)
                    for (int32_t j = 0; j < ml[sg]; ++j) {
                        if ((int32_t)rb.qual(qa[sg] + j) < mq) continue;
                        const uint32_t col = code_to_col(rb.code(qa[sg] + j));
                        if (col == 0xFFu || (uint32_t)(r0[sg] + j) >= (uint32_t)G) { er2 = 1; break; }
                        s2.add(r0[sg] + j, col);
                    }
                if (!er2) er2 = cig2_indels(P, s, t2.pos, lseq, [&](int32_t q) { return (uint32_t)qual[q]; }, s2);
            }
            if ((er1 != 0) != (er2 != 0)) same = false;
            else if (!er1) {
                same = c1 == cc2 && e1.size() == e2.size();
                for (size_t j = 0; same && j < e1.size(); ++j)
                    same = e1[j].ref_pos == e2[j].ref_pos && e1[j].q_from == e2[j].q_from && e1[j].q_to == e2[j].q_to;
            }
        }
        if (!same) {
            if (bad < 8 && getenv("CIG2_DEBUG")) {
                fprintf(stderr, "MISMATCH kind %d m1 %d k %d m2 %d pos %d flag %u tlen %d w %d mq %d | generic err %d pos %d flags %u n %d:", kind, m1, k, m2, pos, flag, tlen, window, mq, st.err, st.pos, st.flags, st.n);
                for (int j = 0; j < st.n; ++j) fprintf(stderr, " %u%c", cur.get(j) >> 4, "MIDNSHP=X"[cur.get(j) & 15]);
                fprintf(stderr, " | cig2 err %d pos %d flags %u n %d:", t2.err, t2.pos, t2.flags, n2);
                for (int j = 0; j < n2; ++j) fprintf(stderr, " %u%c", c2[j] >> 4, "MIDNSHP=X"[c2[j] & 15]);
                fprintf(stderr, " L %d R %d\n", mx[rs], mn[re1 < G && re1 >= 0 ? re1 : 0]);
            }
            ++bad;
        }
    }
    return bad;
}


// ---- fuzz of the branch-free closed forms (amp_bf.hpp) against the branchy ones (Cig2, themselves fuzzed against the
// generic code above): same inputs, same primer tables, every scan outcome i in [0, qlen] ------------------------------
extern "C" long sim_bf_fuzz(uint64_t seed, long iters, long *n_punt, long *n_cmp) {
    rng_state = seed * 0x9E3779B97F4A7C15ull + 777;
    long bad = 0; *n_punt = 0; *n_cmp = 0;
    for (long it = 0; it < iters; ++it) {
        const int kind = rin(0, 2);
        const int32_t m1 = rin(1, 60), k = kind ? rin(1, 20) : 0, m2 = kind ? rin(1, 60) : 0;
        const uint32_t op = (rnd() % 8 == 0) ? OP_EQ : OP_M;
        const int32_t sa = rnd() % 3 == 0 ? rin(1, 30) : 0, sc = rnd() % 3 == 0 ? rin(1, 30) : 0;
        uint32_t in[5] = {0, 0, 0, 0, 0}; int n = 0;
        if (sa) in[n++] = ((uint32_t)sa << 4) | OP_S;
        in[n++] = ((uint32_t)m1 << 4) | op;
        if (kind) { in[n++] = ((uint32_t)k << 4) | (kind == 1 ? OP_I : OP_D); in[n++] = ((uint32_t)m2 << 4) | op; }
        if (sc) in[n++] = ((uint32_t)sc << 4) | OP_S;
        // words past the read's own repeat its first (as the kernel loads them); sometimes a broken CIGAR
        uint32_t w[5];
        for (int j = 0; j < 5; ++j) w[j] = j < n ? in[j] : in[0];
        if (rnd() % 16 == 0) w[rnd() % n] ^= (1u << (rnd() % 8));
        int32_t lseq = sa + m1 + (kind == 1 ? k : 0) + m2 + sc;
        if (rnd() % 32 == 0) lseq += rin(-2, 2);
        const int32_t pos = rin(0, 420);
        const uint32_t flag = (rnd() & 1u) | ((rnd() & 1u) << 4);
        const bool isize = rnd() & 1u;
        // table entries anywhere around the read, or None
        int32_t L = rnd() % 3 == 0 ? -1 : pos + rin(-40, 160), R = rnd() % 3 == 0 ? -1 : pos + rin(-40, 200);
        if (kind && rnd() % 4 == 0) {       // aimed at the indel: a start clip that ends in or next to it, an end clip that reaches it from the other side
            L = pos + m1 + rin(-2, (kind == 2 ? k : 0) + 2) - 1;
            if (rnd() & 1u) R = pos + m1 + (kind == 2 ? k : 0) + rin(-3, 3);
        }
        Cig2 s2;
        const bool ok2 = cig2_from_words5(n, w, lseq, s2) && !((s2.kind == 1 && s2.k > 8) || (s2.kind == 2 && s2.k > 16));
        bool ok3;
        Bf s3 = bf_from_words5(n, w[0], w[1], w[2], w[3], w[4], lseq, 8, 16, ok3);
        if (ok2 != ok3) { ++bad; continue; }
        if (!ok2) continue;
        TrimState t2{pos, n, 0u, 0};
        cig2_trim_primers_isize(t2, flag, isize, lseq, s2, L < 0 ? -1 : L, R < 0 ? -1 : R);
        int32_t p3 = pos; uint32_t f3 = 0u;
        s3 = bf_trim_primers(s3, p3, f3, flag, isize, lseq, L < 0 ? -1 : L, R < 0 ? -1 : R);
        if ((s2.punt ? 1u : 0u) != (s3.punt ? 1u : 0u)) { ++bad; continue; }
        if (s2.punt) { ++*n_punt; continue; }
        auto same_shape = [&](const Cig2 &x, const Bf &y) {
            return x.a == y.a && x.m1 == y.m1 && x.k == y.k && x.m2 == y.m2 && x.c == y.c && x.kind == y.kind && x.op == y.op;
        };
        if (!same_shape(s2, s3) || t2.pos != p3 || t2.flags != f3) { ++bad; continue; }
        int32_t lo2, ql2, lo3, ql3;
        cig2_quality_window(s2, lseq, lo2, ql2);
        bf_quality_window(s3, lseq, lo3, ql3);
        if (lo2 != lo3 || ql2 != ql3) { ++bad; continue; }
        const bool rev = (flag & 0x10u) != 0;
        for (int32_t i = 0; i <= ql2; i += (ql2 > 24 && i > 4 && i < ql2 - 4) ? rin(1, 5) : 1) {
            Cig2 a2 = s2; TrimState u2 = t2;
            cig2_trim_quality(u2, rev, i, ql2, a2);
            uint32_t g3 = f3;
            const Bf a3 = bf_trim_quality(s3, p3, g3, rev, i, ql3);
            ++*n_cmp;
            if ((a2.punt ? 1u : 0u) != (a3.punt ? 1u : 0u)) { ++bad; continue; }
            if (a2.punt) continue;
            if (!same_shape(a2, a3) || u2.flags != g3 || a2.ref_len() != a3.ref_len()) ++bad;
        }
    }
    return bad;
}


// ---- fuzz of the op-by-op indel walk (count_regular_ops, what the tile kernel's lanes run) against the exact pair walk
// (count_read_walk) on random REGULAR CIGARs: H* S* (M|=|X|I|D|N)* S* H*, zero-length ops, insertions next to each other,
// next to deletions, at either end of the body, at reference position 0 and hanging over the reference's end; qualities
// with low bases inside the insertions.  The '-' counts, the events (position and slice) in order and the error / no error
// outcome must be the same. -------------------------------------------------------------------------------------------
struct RecSink {
    std::vector<int32_t> *v;
    void add(int32_t r, uint32_t col) { if (col == 5u) { v->push_back(-1); v->push_back(r); } }
    void event(int32_t pos, int32_t lo, int32_t hi) { v->push_back(pos); v->push_back(lo); v->push_back(hi); }
};
struct QualVec {
    const uint8_t *q;
    uint32_t operator()(int32_t k) const { return q[k]; }
};
struct BytesVec {
    const uint8_t *q;
    uint32_t qual(int32_t i) { return q[i]; }
    uint32_t code(int32_t) { return 1u; }
};

extern "C" long sim_regops_fuzz(uint64_t seed, long iters, long *n_events, long *n_errors) {
    rng_state = seed * 0x9E3779B97F4A7C15ull + 4242;
    long bad = 0; *n_events = 0; *n_errors = 0;
    const int32_t G = 600;
    KParams P{20, 4, 1, 1, G, 0, nullptr, nullptr};
    std::vector<uint32_t> cig;
    std::vector<uint8_t> qual;
    std::vector<int32_t> e1, e2;
    for (long it = 0; it < iters; ++it) {
        cig.clear();
        if (rnd() % 8 == 0) cig.push_back(((uint32_t)rin(0, 5) << 4) | OP_H);
        if (rnd() % 3 == 0) cig.push_back(((uint32_t)rin(0, 20) << 4) | OP_S);
        const int nbody = rin(1, 9);
        int32_t qlen = 0, rlen = 0;
        for (int k = 0; k < nbody; ++k) {
            const int t = (int)(rnd() % 10);
            uint32_t op = t < 4 ? OP_M : t < 6 ? OP_I : t < 8 ? OP_D : t == 8 ? OP_N : (rnd() & 1u ? OP_EQ : OP_X);
            int32_t len = rnd() % 12 == 0 ? 0 : rin(1, 14);
            cig.push_back(((uint32_t)len << 4) | op);
        }
        if (rnd() % 3 == 0) cig.push_back(((uint32_t)rin(0, 20) << 4) | OP_S);
        if (rnd() % 8 == 0) cig.push_back(((uint32_t)rin(0, 5) << 4) | OP_H);
        for (uint32_t v : cig) { const uint32_t op = v & 15u; if (consumes_query(op)) qlen += (int32_t)(v >> 4); if (consumes_ref(op)) rlen += (int32_t)(v >> 4); }
        if (qlen == 0) continue;
        const int n = (int)cig.size();
        const int32_t lseq = qlen;
        qual.resize((size_t)lseq);
        const int lowp = (int)(rnd() % 3);          // no / some / many low bases
        for (int32_t i = 0; i < lseq; ++i) qual[(size_t)i] = (uint8_t)((lowp && (int)(rnd() % (lowp == 1 ? 8 : 2)) == 0) ? rin(0, 19) : rin(20, 40));
        int32_t pos = rnd() % 4 == 0 ? 0 : rin(0, G - 1);
        const bool inside = pos + rlen <= G;
        if (!inside && rnd() % 4) pos = rin(0, G - rlen > 0 ? G - rlen : 0);
        const bool in2 = pos + (rlen ? rlen : 1) <= G;
        CigBuf<1> cb{cig.data()};
        int er = 0;
        const int32_t qs = query_alignment_start(cb, n, lseq, er), qe = query_alignment_end(cb, n, lseq, er);
        if (er) continue;
        e1.clear(); e2.clear();
        RecSink s1{&e1}, s2{&e2};
        const int rc1 = count_read_walk(P, cb, n, pos, lseq, BytesVec{qual.data()}, true, s1);
        const int rc2 = count_regular_ops(P, cb, n, pos, pos + reference_length(cb, n), lseq, qs, qe, QualVec{qual.data()}, s2);
        *n_events += (long)e2.size() / 3;
        if (rc1) ++*n_errors;
        if (e1 != e2) { ++bad; continue; }
        if (in2 && (rc1 != 0) != (rc2 != 0)) ++bad;
    }
    return bad;
}
