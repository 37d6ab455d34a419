// amp_ins.hip -- on-device aggregation of insertion events.
//
// The reference tallies insertion alleles in a dict keyed by the allele string (AmpliPy.py:745-748) and ranks them
// with the base symbols (A:767-771).  The kernels record an insertion as the integer event (ref_pos, read, q_from, q_to)
// -- the allele is SEQ[q_from:q_to] of that read (A:736-738) -- and until now the host built the tally from the event
// list.  Here the device does it: every event gets the key (ref_pos, length, 64-bit hash of its 4-bit base codes), the
// keys are radix-sorted (rocPRIM through hipcub: two stable passes, hash first, then position | length), runs of equal
// keys are closed by comparing the base codes of neighbours exactly (a hash collision splits a run, it never merges two
// alleles), and one record per run -- a representative event and the number of events -- goes back to the host.
// The 4-bit codes map one to one onto the letters "=ACMGRSVTWYHKDBN" (A:702 upper-cases the read), so equal codes
// <=> equal allele text.
//
// A translation unit of its own: the sort's headers triple the compile time of whatever includes them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <stdint.h>

#include "amp_ins.hpp"

namespace amp {

struct InsLayout {
    uint64_t *key_a, *key_b;      // sort keys, ping-pong
    uint32_t *idx_a, *idx_b;      // event numbers, ping-pong
    uint32_t *head;               // 1 at the first event of a run (sorted order); then its exclusive prefix sum
    uint32_t *rid;
    unsigned long long *nvalid;   // [0] events that are real (slots reserved and not used carry ref_pos = -1), [1] runs
    void *tmp; size_t tmp_bytes;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_tmp_bytes(int64_t n) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    return a > b ? a : b;
}

size_t ins_scratch_bytes(int64_t n_slots) {
    const size_t n = (size_t)(n_slots > 0 ? n_slots : 1);
    return 2 * align_up(n * 8) + 4 * align_up(n * 4) + 256 + align_up(sort_tmp_bytes((int64_t)n));
}

static InsLayout carve(void *scratch, int64_t n_slots) {
    const size_t n = (size_t)(n_slots > 0 ? n_slots : 1);
    uint8_t *p = (uint8_t *)scratch;
    InsLayout L;
    L.key_a = (uint64_t *)p; p += align_up(n * 8);
    L.key_b = (uint64_t *)p; p += align_up(n * 8);
    L.idx_a = (uint32_t *)p; p += align_up(n * 4);
    L.idx_b = (uint32_t *)p; p += align_up(n * 4);
    L.head = (uint32_t *)p; p += align_up(n * 4);
    L.rid = (uint32_t *)p; p += align_up(n * 4);
    L.nvalid = (unsigned long long *)p; p += 256;
    L.tmp = p; L.tmp_bytes = align_up(sort_tmp_bytes((int64_t)n));
    return L;
}

__device__ __forceinline__ uint32_t ins_code(const amp_dev_reads &rd, int64_t boff, int32_t q) {
    const int64_t k = boff + q;
    const uint32_t b = rd.seq[k >> 1];
    return (k & 1) ? (b & 15u) : (b >> 4);
}

// slot j of the concatenated shard regions -> its event
struct ShardMap { const amp_ins_event *ev; long long cap; long long start[9]; };      // start[s] = slots in front of shard s
__device__ __forceinline__ const amp_ins_event &slot_event(const ShardMap &M, int64_t j) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += j >= M.start[k] ? 1 : 0;
    return M.ev[(size_t)s * (size_t)M.cap + (size_t)(j - M.start[s])];
}

// pass 1: hash of the allele's base codes (the low sort key); unused slots sort behind everything
__global__ void k_ins_hash(amp_dev_reads rd, uint64_t read_base, ShardMap M, int64_t n_slots, uint64_t *key, uint32_t *idx, unsigned long long *nvalid) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_slots) return;
    const amp_ins_event e = slot_event(M, j);
    idx[j] = (uint32_t)j;
    if (e.ref_pos < 0) { key[j] = ~0ull; return; }
    const int64_t i = (int64_t)((uint64_t)e.read - read_base) & 0xFFFFFFFFll;
    const int64_t boff = (int64_t)rd.seq_off8[i] * 8;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int32_t q = e.q_from; q < e.q_to; ++q) { h ^= (uint64_t)ins_code(rd, boff, q) + 1ull; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
    key[j] = h >> 1;                                         // (the top bit is kept for the unused slots)
    atomicAdd(&nvalid[0], 1ull);
}

// pass 2: the high sort key of the events in hash order: ref_pos << 32 | length
__global__ void k_ins_poskey(ShardMap M, int64_t n_slots, const uint32_t *idx, uint64_t *key) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_slots) return;
    const amp_ins_event e = slot_event(M, (int64_t)idx[j]);
    key[j] = e.ref_pos < 0 ? ~0ull : ((uint64_t)(uint32_t)e.ref_pos << 32) | (uint64_t)(uint32_t)(e.q_to - e.q_from);
}

// an event opens a run when it differs from its predecessor in position, length or any base code
__global__ void k_ins_heads(amp_dev_reads rd, uint64_t read_base, ShardMap M, const unsigned long long *nvalid, const uint64_t *key, const uint32_t *idx,
                            uint32_t *head) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)nvalid[0];
    if (j >= n) return;
    uint32_t h = 1u;
    if (j > 0 && key[j] == key[j - 1]) {
        const amp_ins_event a = slot_event(M, (int64_t)idx[j]), b = slot_event(M, (int64_t)idx[j - 1]);
        const int64_t ia = (int64_t)((uint64_t)a.read - read_base) & 0xFFFFFFFFll, ib = (int64_t)((uint64_t)b.read - read_base) & 0xFFFFFFFFll;
        const int64_t oa = (int64_t)rd.seq_off8[ia] * 8, ob = (int64_t)rd.seq_off8[ib] * 8;
        const int32_t len = a.q_to - a.q_from;
        bool same = true;
        for (int32_t q = 0; q < len && same; ++q) same = ins_code(rd, oa, a.q_from + q) == ins_code(rd, ob, b.q_from + q);
        h = same ? 0u : 1u;
    }
    head[j] = h;
}

// one record per run: the head writes the representative, every event adds one to its run's count
__global__ void k_ins_runs(ShardMap M, const unsigned long long *nvalid, const uint32_t *idx, const uint32_t *head, const uint32_t *rid, amp_ins_run *runs,
                           unsigned long long *nruns) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)nvalid[0];
    if (j >= n) return;
    // rid = exclusive sum of head: the run of event j is rid[j] + head[j] - 1
    const uint32_t r = rid[j] + head[j] - 1u;
    if (head[j]) runs[r].first = slot_event(M, (int64_t)idx[j]);
    atomicAdd(&runs[r].count, 1u);
    if (j == n - 1) nruns[0] = (unsigned long long)r + 1ull;
}

int ins_aggregate(hipStream_t s, const amp_dev_reads &rd, uint64_t read_base, const amp_ins_event *ev, long long cap, const unsigned long long *shard_n,
                  void *scratch, amp_ins_run *d_runs, int64_t *n_events, int64_t *n_runs) {
    ShardMap M;
    M.ev = ev; M.cap = cap;
    long long tot = 0;
    for (int k = 0; k < 8; ++k) { M.start[k] = tot; tot += (long long)shard_n[k]; }
    M.start[8] = tot;
    *n_events = 0; *n_runs = 0;
    if (tot == 0) return 0;
    if (tot > 0x7FFFFFFFll) return -1;
    const int64_t n = tot;
    InsLayout L = carve(scratch, n);
    const unsigned g = (unsigned)((n + 255) / 256);
    hipError_t e = hipMemsetAsync(L.nvalid, 0, 16, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(d_runs, 0, (size_t)n * sizeof(amp_ins_run), s);
    if (e != hipSuccess) return (int)e;
    k_ins_hash<<<g, 256, 0, s>>>(rd, read_base, M, n, L.key_a, L.idx_a, L.nvalid);
    size_t tb = L.tmp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(L.tmp, tb, L.key_a, L.key_b, L.idx_a, L.idx_b, (int)n, 0, 64, s);
    if (e != hipSuccess) return (int)e;
    k_ins_poskey<<<g, 256, 0, s>>>(M, n, L.idx_b, L.key_a);
    tb = L.tmp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(L.tmp, tb, L.key_a, L.key_b, L.idx_b, L.idx_a, (int)n, 0, 64, s);     // (stable: equal keys keep the hash order)
    if (e != hipSuccess) return (int)e;
    // sorted: keys in key_b, event numbers in idx_a.  Events with the same allele are neighbours unless two different
    // alleles of one position and length collide in the hash AND interleave -- then an allele shows up as more than one run
    // (the consumer sums runs by text); k_ins_heads compares the codes, so no run ever mixes two alleles.
    k_ins_heads<<<g, 256, 0, s>>>(rd, read_base, M, L.nvalid, L.key_b, L.idx_a, L.head);
    tb = L.tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(L.tmp, tb, L.head, L.rid, (int)n, s);
    if (e != hipSuccess) return (int)e;
    k_ins_runs<<<g, 256, 0, s>>>(M, L.nvalid, L.idx_a, L.head, L.rid, d_runs, L.nvalid + 1);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    unsigned long long h[2] = {0, 0};
    e = hipMemcpyAsync(h, L.nvalid, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    *n_events = (int64_t)h[0];
    *n_runs = h[0] ? (int64_t)h[1] : 0;
    return 0;
}

}  // namespace amp
