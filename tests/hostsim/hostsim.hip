// hostsim -- TEST INFRASTRUCTURE.  Runs the per-read device functions of
// amplipy_amd/csrc/amp_read.hpp on the CPU so that their logic can be checked against the
// golden vectors in a container without a GPU.  Never loaded by the amplipy_amd package
// and not a fallback: the product library has no host execution path.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../amplipy_amd/csrc/amp_read.hpp"

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
