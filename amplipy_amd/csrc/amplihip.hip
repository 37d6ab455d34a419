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
#include <new>
#include <vector>

#include "../../include/amplihip.h"
#include "amp_read.hpp"
#include "amp_tile.hpp"

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
    DBuf events;                  // amp_ins_event[ev_cap]
    int64_t ev_cap = 0;
    bool ev_reserved = false;     // caller sized the buffer: skip the bound pre-pass
    unsigned long long *d_ctr = nullptr;  // [0] events recorded, [1] event bound, [2] error-read count
    // staging for the host-pointer path
    DBuf s_pos, s_flag, s_tlen, s_lseq, s_cigoff, s_cig, s_seqoff, s_seq, s_qual;
    DBuf o_pos, o_ncig, o_cig, o_reflen, o_flags, o_status;
    DBuf scratch;                 // CIGAR scratch for reads whose ops do not fit the LDS slots
    DBuf call_buf;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    bool timed = false;
    int n_cu = 256;
    int kernel_variant = 2;       // 1 = one lane per read (reference kernels), 2 = tile kernel
    uint32_t phases = 0xFFFFFFFFu; // debug: phases of the tile kernel to run (AMPLIHIP_PHASES)
    char err[320] = {0};
};

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,               \
                     hipGetErrorString(e__), __FILE__, __LINE__);                                  \
            return e__ == hipErrorOutOfMemory ? AMP_ENOMEM : AMP_EHIP;                             \
        }                                                                                          \
    } while (0)

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
    amp_ins_event *ev;
    unsigned long long *ctr;
    long long ev_cap;
    uint32_t read;
    __device__ void add(int32_t r, uint32_t col) { atomicAdd(&counts[(size_t)r * AMP_NSYM + col], 1u); }
    __device__ void event(int32_t pos, int32_t lo, int32_t hi) {
        unsigned long long idx = atomicAdd(&ctr[0], 1ull);
        if ((long long)idx < ev_cap) ev[idx] = amp_ins_event{pos, read, lo, hi};
    }
};


struct NullSink {   // dry run: only the status matters
    __device__ void add(int32_t, uint32_t) {}
    __device__ void event(int32_t, int32_t, int32_t) {}
};

// One read, start to finish, on one lane with the serial code of amp_read.hpp (CIGAR ping-pong
// in global memory).  status_only: the tile kernel already counted this read and only needs
// to know which error comes first in pair order.
__device__ void process_read_serial(const KParams &P, const amp_dev_reads &rd, int64_t i, uint64_t read_base, const DevOut &out,
                                    uint32_t *scratch, uint32_t *counts, amp_ins_event *ev, unsigned long long *ctr,
                                    long long ev_cap, bool status_only) {
    const uint32_t c0 = rd.cig_off32[i];
    const int n = (int)(rd.cig_off32[i + 1] - c0);
    const size_t slot = (size_t)c0 + 3 * (size_t)i;
    CigBuf<1> cur{out.new_cig + slot}, tmp{scratch + slot};
    uint32_t *const home = cur.p;
    for (int k = 0; k < n; ++k) cur.set(k, rd.cig[c0 + k]);
    const int32_t lseq = (int32_t)rd.lseq[i];
    const int64_t boff = (int64_t)rd.seq_off8[i] * 8;
    const uint8_t *qual = rd.qual + boff;
    const bool have_qual = lseq > 0 && qual[0] != 0xFF;
    TrimState st{rd.pos[i], n, 0u, 0};
    if (P.do_trim) trim_read_serial(P, st, rd.flag[i], rd.tlen[i], lseq, qual, have_qual, cur, tmp);
    if (!st.err && cur.p != home) {
        for (int k = 0; k < st.n; ++k) home[k] = cur.get(k);
        cur.p = home;
    }
    int err = st.err;
    if (!err && P.do_count) {
        if (status_only) {
            NullSink ns;
            err = count_read_walk(P, cur, st.n, st.pos, lseq, rd.seq, boff, qual, have_qual, ns);
        } else {
            DevSink sink{counts, ev, ctr, ev_cap, (uint32_t)(read_base + (uint64_t)i)};
            err = count_read_walk(P, cur, st.n, st.pos, lseq, rd.seq, boff, qual, have_qual, sink);
        }
    }
    if (out.new_pos) out.new_pos[i] = st.pos;
    if (out.new_ncig) out.new_ncig[i] = st.err ? 0u : (uint32_t)st.n;
    if (out.ref_len) out.ref_len[i] = st.err ? 0 : reference_length(cur, st.n);
    if (out.trim_flags) out.trim_flags[i] = st.err ? (uint8_t)0 : (uint8_t)st.flags;
    if (out.status) out.status[i] = (uint8_t)err;
    if (err) atomicAdd(&ctr[2], 1ull);
}

// Variant 1: every read on its own lane.  Kept as the simple kernel the tile kernel is
// A/B-checked against on the GPU.
__global__ void __launch_bounds__(256)
k_reads_lane(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *scratch, uint32_t *counts,
             amp_ins_event *ev, unsigned long long *ctr, long long ev_cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rd.n_reads) return;
    process_read_serial(P, rd, i, read_base, out, scratch, counts, ev, ctr, ev_cap, false);
}

// Second pass of variant 2: the reads the tile kernel put on its deferred list.
__global__ void __launch_bounds__(256)
k_reads_deferred(KParams P, amp_dev_reads rd, uint64_t read_base, DevOut out, uint32_t *scratch, uint32_t *counts,
                 amp_ins_event *ev, unsigned long long *ctr, long long ev_cap, const uint32_t *dlist) {
    const unsigned long long cnt = ctr[3];
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < cnt;
         k += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t e = dlist[k];
        process_read_serial(P, rd, (int64_t)(e & 0x7FFFFFFFu), read_base, out, scratch, counts, ev, ctr, ev_cap,
                            (e & DEFER_STATUS_ONLY) != 0);
    }
}

__global__ void k_add_u32(uint32_t *dst, const uint32_t *src, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// Calling, integer part (A:756-771): per position the total of the six base symbols and
// their order under sorted(..., reverse=True): count descending, ties by symbol descending
// ('T' > 'N' > 'G' > 'C' > 'A' > '-').  order = six 3-bit column indices, best first.
__global__ void k_rank_bases(const uint32_t *__restrict__ counts, int32_t ref_len, uint32_t *__restrict__ total6,
                             uint32_t *__restrict__ order) {
    int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ref_len) return;
    // columns in descending symbol order: T(3) N(4) G(2) C(1) A(0) -(5)
    const int desc[6] = {3, 4, 2, 1, 0, 5};
    uint32_t c[6], idx[6];
    uint32_t tot = 0;
    for (int k = 0; k < 6; ++k) { idx[k] = desc[k]; c[k] = counts[(size_t)p * AMP_NSYM + desc[k]]; tot += c[k]; }
    // stable insertion sort by count descending keeps the symbol order for ties
    for (int a = 1; a < 6; ++a) {
        uint32_t cv = c[a], iv = idx[a];
        int b = a - 1;
        while (b >= 0 && c[b] < cv) { c[b + 1] = c[b]; idx[b + 1] = idx[b]; --b; }
        c[b + 1] = cv; idx[b + 1] = iv;
    }
    uint32_t o = 0;
    for (int k = 0; k < 6; ++k) o |= idx[k] << (3 * k);
    total6[p] = tot;
    order[p] = o;
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
    size_t cb = (size_t)ref_len * AMP_NSYM * sizeof(uint32_t);
    if (hipMalloc((void **)&c->d_counts, cb) != hipSuccess) return fail(AMP_ENOMEM);
    c->own_counts = true;
    if (hipMalloc((void **)&c->d_min_start, (size_t)ref_len * 4) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMalloc((void **)&c->d_max_end, (size_t)ref_len * 4) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMalloc((void **)&c->d_ctr, 8 * sizeof(unsigned long long)) != hipSuccess) return fail(AMP_ENOMEM);
    if (hipMemsetAsync(c->d_counts, 0, cb, c->stream) != hipSuccess) return fail(AMP_EHIP);
    if (hipMemsetAsync(c->d_ctr, 0, 8 * sizeof(unsigned long long), c->stream) != hipSuccess) return fail(AMP_EHIP);
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreate(&c->ev2) != hipSuccess || hipEventCreate(&c->ev3) != hipSuccess) return fail(AMP_EHIP);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(AMP_EHIP);
    const char *v = getenv("AMPLIHIP_KERNEL");
    if (v && (v[0] == '1' || v[0] == '2')) c->kernel_variant = v[0] - '0';
    v = getenv("AMPLIHIP_PHASES");
    if (v) c->phases = (uint32_t)strtoul(v, nullptr, 0);
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
    DBuf *bufs[] = {&c->events, &c->s_pos, &c->s_flag, &c->s_tlen, &c->s_lseq, &c->s_cigoff, &c->s_cig, &c->s_seqoff,
                    &c->s_seq, &c->s_qual, &c->o_pos, &c->o_ncig, &c->o_cig, &c->o_reflen, &c->o_flags, &c->o_status,
                    &c->scratch, &c->call_buf};
    for (DBuf *b : bufs) b->release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
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
    size_t cb = (size_t)c->ref_len * AMP_NSYM * sizeof(uint32_t);
    HIPCHK(c, hipMemcpyAsync(dev_counts, c->d_counts, cb, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_counts) (void)hipFree(c->d_counts);
    c->d_counts = (uint32_t *)dev_counts;
    c->own_counts = false;
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

int amp_set_kernel_variant(amp_ctx *c, int variant) {  // 1 = lane-per-read kernels, 2 = tile kernel
    if (!c || (variant != 1 && variant != 2)) return AMP_EINVAL;
    c->kernel_variant = variant;
    return AMP_OK;
}

int amp_reserve_events(amp_ctx *c, int64_t cap) {
    if (!c || cap < 0) return AMP_EINVAL;
    Guard g(c);
    HIPCHK(c, c->events.ensure((size_t)cap * sizeof(amp_ins_event), true, c->stream));
    c->ev_cap = std::max(c->ev_cap, cap);
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
    if (n == 0) return AMP_OK;
    // event capacity
    if (c->do_count && !c->ev_reserved) {
        unsigned long long h[2] = {0, 0};
        HIPCHK(c, hipMemsetAsync(&c->d_ctr[1], 0, sizeof(unsigned long long), c->stream));
        k_event_bound<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(n, rd->cig_off32, rd->cig, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int64_t need = (int64_t)(h[0] + h[1]);
        if (need > c->ev_cap) {
            int64_t ncap = std::max<int64_t>(need, c->ev_cap + c->ev_cap / 2);
            HIPCHK(c, c->events.ensure((size_t)ncap * sizeof(amp_ins_event), true, c->stream));
            c->ev_cap = ncap;
        }
    }
    KParams P{c->min_quality, c->window, c->do_trim, c->do_count, c->ref_len, c->max_primer_len, c->d_min_start, c->d_max_end};
    if (n > 0x7FFFFFFFll) return AMP_EINVAL;
    const size_t slots = (size_t)rd->n_cig + 3 * (size_t)n;
    DevOut out{o ? o->new_pos : nullptr, o ? o->new_ncig : nullptr, o ? o->new_cig : nullptr, o ? o->ref_len : nullptr,
               o ? o->trim_flags : nullptr, o ? o->status : nullptr};
    // scratch: [CIGAR ping-pong slots][deferred list][trimmed CIGARs when the caller does not want them]
    HIPCHK(c, c->scratch.ensure((slots * (out.new_cig ? 1 : 2) + (size_t)n) * 4));
    uint32_t *scr = c->scratch.as<uint32_t>();
    uint32_t *dlist = scr + slots;
    if (!out.new_cig) out.new_cig = dlist + n;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->kernel_variant == 1) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        k_reads_lane<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(P, *rd, read_base, out, scr, c->d_counts,
                                                                        c->events.as<amp_ins_event>(), c->d_ctr, (long long)c->ev_cap);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev2, c->stream));
    } else {
        HIPCHK(c, hipMemsetAsync(&c->d_ctr[3], 0, sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        int rc = tile_launch(P, *rd, read_base, out, c->d_counts, c->events.as<amp_ins_event>(), c->d_ctr,
                             (long long)c->ev_cap, dlist, c->n_cu, c->phases, c->stream);
        if (rc != 0) { snprintf(c->err, sizeof(c->err), "tile kernel launch failed: %s", hipGetErrorString((hipError_t)rc)); return AMP_EHIP; }
        HIPCHK(c, hipEventRecord(c->ev2, c->stream));
        k_reads_deferred<<<(unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)c->n_cu * 4), 256, 0, c->stream>>>(
            P, *rd, read_base, out, scr, c->d_counts, c->events.as<amp_ins_event>(), c->d_ctr, (long long)c->ev_cap, dlist);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev3, c->stream));
    c->timed = true;
    return AMP_OK;
}

int amp_process_batch_device(amp_ctx *c, const amp_dev_reads *rd, uint64_t read_base, const amp_trim_out *dev_out) {
    if (!c || !rd || rd->n_reads < 0) return AMP_EINVAL;
    if (rd->n_reads && (!rd->pos || !rd->flag || !rd->tlen || !rd->lseq || !rd->cig_off32 || !rd->cig || !rd->seq_off8 ||
                        !rd->seq || !rd->qual)) return AMP_EINVAL;
    Guard g(c);
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
    HIPCHK(c, hipEventElapsedTime(&s, c->ev1, c->ev2));
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
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n = (int64_t)h;
    if ((int64_t)h > c->ev_cap) return AMP_EOVERFLOW;  // reserved buffer was too small: events were dropped
    if (buf) {
        int64_t m = std::min<int64_t>((int64_t)h, cap);
        if (m > 0) {
            HIPCHK(c, hipMemcpyAsync(buf, c->events.p, (size_t)m * sizeof(amp_ins_event), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if ((int64_t)h > cap) return AMP_EOVERFLOW;
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
    HIPCHK(c, hipMemsetAsync(c->d_counts, 0, (size_t)c->ref_len * AMP_NSYM * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, 8 * sizeof(unsigned long long), c->stream));
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
    void *h = dlopen(nullptr, RTLD_NOW);
    void *f_red = h ? dlsym(h, "ncclReduce") : nullptr;
    void *f_all = h ? dlsym(h, "ncclAllReduce") : nullptr;
    if (!f_red || !f_all) {
        void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) { f_red = dlsym(lib, "ncclReduce"); f_all = dlsym(lib, "ncclAllReduce"); }
    }
    if (!f_red || !f_all) { snprintf(c->err, sizeof(c->err), "RCCL symbols not found"); return AMP_ERCCL; }
    const size_t cnt = (size_t)c->ref_len * AMP_NSYM;
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
static const char SYMS[6] = {'A', 'C', 'G', 'T', 'N', '-'};
// Number of base symbols that sort AFTER an insertion string s in descending order is decided
// by s's first character: compare s with each one-character symbol as Python strings.
static inline bool str_gt_sym(const uint8_t *s, int32_t len, char sym) {  // s > sym ?
    if (len == 0) return false;                 // '' < anything
    if ((char)s[0] != sym) return (unsigned char)s[0] > (unsigned char)sym;
    return len > 1;                             // "Ax" > "A"
}

int amp_call(amp_ctx *c, const amp_call_params *pr, const uint8_t *ref_seq, int64_t n_ins, const int32_t *ins_pos,
             const uint32_t *ins_count, const uint8_t *const *ins_str, const int32_t *ins_len, const amp_call_out *out) {
    if (!c || !pr || !out || n_ins < 0 || (n_ins && (!ins_pos || !ins_count || !ins_str || !ins_len))) return AMP_EINVAL;
    if (pr->run_variants && !ref_seq) return AMP_EINVAL;
    Guard g(c);
    const int32_t G = c->ref_len;
    HIPCHK(c, c->call_buf.ensure((size_t)G * 8));
    uint32_t *d_tot = c->call_buf.as<uint32_t>(), *d_ord = d_tot + G;
    k_rank_bases<<<(unsigned)((G + 255) / 256), 256, 0, c->stream>>>(c->d_counts, G, d_tot, d_ord);
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> tot((size_t)G), ord((size_t)G), cnt((size_t)G * AMP_NSYM);
    HIPCHK(c, hipMemcpyAsync(tot.data(), d_tot, (size_t)G * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ord.data(), d_ord, (size_t)G * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cnt.data(), c->d_counts, (size_t)G * AMP_NSYM * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t k = 0, na = 0;
    for (int32_t p = 0; p < G; ++p) {
        const int64_t k0 = k;
        uint64_t total = tot[(size_t)p];
        while (k < n_ins && ins_pos[k] == p) { total += ins_count[k]; ++k; }
        if (k < n_ins && ins_pos[k] < p) return AMP_EINVAL;  // rows must be sorted by position
        const int nins = (int)(k - k0);
        // merge: base symbols (ranked on the device) with this position's insertion rows
        amp_allele al[6];
        int nb = 0;
        for (int r = 0; r < 6; ++r) {
            int col = (ord[(size_t)p] >> (3 * r)) & 7;
            uint32_t cv = cnt[(size_t)p * AMP_NSYM + col];
            if (cv) al[nb++] = amp_allele{cv, col};
        }
        const int64_t a0 = na;
        if (out->allele_off) out->allele_off[p] = (uint64_t)a0;
        int n_all = nb;
        std::vector<amp_allele> merged;
        const amp_allele *ranked = al;
        if (nins) {
            merged.reserve((size_t)nb + nins);
            std::vector<int> rows;
            for (int j = 0; j < nins; ++j) if (ins_count[k0 + j]) rows.push_back(j);
            // rows arrive in descending string order; stable sort by count keeps it for ties
            std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return ins_count[k0 + a] > ins_count[k0 + b]; });
            size_t ib = 0, ii = 0;
            while (ib < (size_t)nb || ii < rows.size()) {
                bool take_ins;
                if (ib >= (size_t)nb) take_ins = true;
                else if (ii >= rows.size()) take_ins = false;
                else {
                    uint32_t cb = al[ib].count, ci = ins_count[k0 + rows[ii]];
                    if (ci != cb) take_ins = ci > cb;
                    else take_ins = str_gt_sym(ins_str[k0 + rows[ii]], ins_len[k0 + rows[ii]], SYMS[al[ib].sym]);
                }
                if (take_ins) { merged.push_back(amp_allele{ins_count[k0 + rows[ii]], 6 + rows[ii]}); ++ii; }
                else merged.push_back(al[ib++]);
            }
            ranked = merged.data();
            n_all = (int)merged.size();
        }
        if (out->total_depth) out->total_depth[p] = (uint32_t)total;
        if (out->n_alleles) out->n_alleles[p] = n_all;
        if (out->alleles) {
            if (na + n_all > out->alleles_cap) return AMP_EOVERFLOW;
            for (int r = 0; r < n_all; ++r) out->alleles[na + r] = ranked[r];
        }
        na += n_all;
        // consensus (A:928-929)
        if (out->consensus_sym) {
            int32_t cs = -1;
            if (pr->run_consensus && n_all && ranked[0].count >= (uint32_t)std::max(pr->min_depth_consensus, 0) &&
                (double)ranked[0].count / (double)total >= pr->min_freq_consensus) cs = ranked[0].sym;
            out->consensus_sym[p] = cs;
        }
        // variants (A:933-951)
        if (pr->run_variants) {
            const char ref_symbol = (char)ref_seq[p];
            uint64_t tot_count = 0;
            uint32_t rc = 0; double rf = 0; int n_alt = 0;
            for (int r = 0; r < n_all; ++r) {
                tot_count += ranked[r].count;
                double f = (double)ranked[r].count / (double)total;
                bool is_ref = ranked[r].sym < 6 ? SYMS[ranked[r].sym] == ref_symbol
                                                : (ins_len[k0 + ranked[r].sym - 6] == 1 && (char)ins_str[k0 + ranked[r].sym - 6][0] == ref_symbol);
                uint8_t af = 0;
                if (is_ref) { rc = ranked[r].count; rf = f; af = 2; }
                else if (f >= pr->min_freq_variants) { af = 1; ++n_alt; }
                if (out->allele_flags && out->alleles) out->allele_flags[a0 + r] = af;
            }
            uint8_t vf = 0;
            if ((int64_t)tot_count >= (int64_t)pr->min_depth_variants && n_alt) {
                vf |= 1;
                if ((int64_t)rc >= (int64_t)pr->min_depth_variants && rf >= pr->min_freq_variants) vf |= 2;
            }
            if (out->variant_flags) out->variant_flags[p] = vf;
            if (out->ref_count) out->ref_count[p] = rc;
        } else {
            if (out->variant_flags) out->variant_flags[p] = 0;
        }
    }
    if (out->allele_off) out->allele_off[G] = (uint64_t)na;
    return AMP_OK;
}

}  // extern "C"
