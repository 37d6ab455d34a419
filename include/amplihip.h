/*
 * amplihip.h -- C ABI of libamplihip.so, the MI355X (gfx950) engine behind AmpliPy's
 * trim + pileup + call path.
 *
 * AmpliPy (the reference) has no FFI of its own: its hot path is three Python function
 * seams called once per read from run_amplipy's loop (AmpliPy.py:896-915):
 *     trim_read(s, min_primer_start, max_primer_end, max_primer_len, min_quality, window)
 *                                                         AmpliPy.py:426-687, called :907
 *     update_base_counts(symbol_counts_at_ref_pos, s, min_quality)
 *                                                         AmpliPy.py:690-753, called :915
 *     alleles_from_counts(symbol_counts) + calling loop   AmpliPy.py:756-771, :917-952
 * plus find_overlapping_primers (AmpliPy.py:174-209) which builds trim_read's two tables.
 * A per-read FFI call would cost more than the work, so the replacement is batch-level:
 * one call takes a packed structure-of-arrays batch of reads and performs, for every read
 * in order, exactly what :907 and :915 do.  Each entry point below names the reference
 * lines it replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds to AmpliPy.py.
 *
 * Conventions
 *   - every function returns AMP_OK (0) or a negative amp_rc; nothing throws or aborts
 *   - the caller owns every buffer it passes; a ctx owns its device memory
 *   - one ctx per device; a ctx is not thread-safe; different ctxs are independent
 *   - there is NO CPU fallback: amp_ctx_create fails with AMP_ENODEV without a GPU
 *   - all integers little-endian, arrays contiguous
 */
#ifndef AMPLIHIP_H
#define AMPLIHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMP_ABI_VERSION 1
#define AMP_NSYM 6 /* count-table columns, in this order: A C G T N '-'  (AmpliPy.py:892) */
#define AMP_SEQ_ALIGN 8 /* every read starts on a multiple of 8 bases in seq/qual */
#define AMP_DEV_COLS 7 /* device table: uint32[ref_len][6] counts, then uint32[ref_len] insertion-event tally */

typedef enum amp_rc {
    AMP_OK = 0,
    AMP_EINVAL = -1,    /* bad argument (null pointer, negative size, window < 1 ...) */
    AMP_ENOMEM = -2,    /* host or device allocation failed */
    AMP_EHIP = -3,      /* a HIP runtime call failed (see amp_last_error) */
    AMP_ENODEV = -4,    /* no usable gfx950 device */
    AMP_ESTATE = -5,    /* call order violated (e.g. trim requested before amp_set_primers) */
    AMP_EOVERFLOW = -6, /* an output buffer supplied by the caller is too small */
    AMP_ERCCL = -7      /* RCCL not available / collective failed */
} amp_rc;

/*
 * Per-read status: the inputs on which the reference raises.  The
 * engine reports the condition instead of a result; amp_read_status_exception() names the
 * Python exception class AmpliPy raises for it.  When any read of a batch has a non-zero
 * status the batch's count contribution is unspecified (the reference would have died).
 */
typedef enum amp_read_status {
    AMP_RS_OK = 0,
    AMP_RS_INDEX_REF = 1,   /* IndexError: reference coordinate outside [0, ref_len)   :450-451, :715, :745, :753 */
    AMP_RS_INDEX_PAIRS = 2, /* IndexError: insertion run reaches the end of the pairs   :734 */
    AMP_RS_INDEX_QUERY = 3, /* IndexError: query index beyond the stored SEQ/QUAL       :718 */
    AMP_RS_KEY_BASE = 4,    /* KeyError: counted base is not one of A C G T N           :753 */
    AMP_RS_NO_SEQ = 5,      /* AttributeError: SEQ is '*'                               :702 */
    AMP_RS_NO_QUAL = 6,     /* TypeError: QUAL is '*'                                   :561-562, :718 */
    AMP_RS_CLIP = 7,        /* ValueError: hard clip inside soft clip (pysam accessor)  :561, :700-701 */
    AMP_RS_CIGAR_OP = 8,    /* IndexError: CIGAR op code >= 9                           :378, :404, :474 */
    AMP_RS_TYPE = 9         /* TypeError: None arithmetic (insertion followed by a deletion at ref 0) :736 */
} amp_read_status;

/* bits of trim_flags[i]: trim_read's three return values (AmpliPy.py:687) */
#define AMP_TRIM_PRIMER_START 1u
#define AMP_TRIM_PRIMER_END 2u
#define AMP_TRIM_QUALITY 4u

typedef struct amp_ctx amp_ctx;

/* One insertion allele observation (AmpliPy.py:730-748): the string is
 * SEQ[q_from:q_to] of read `read` (bounds already resolved, 0 <= q_from <= q_to <= l_seq),
 * counted at reference position ref_pos. */
typedef struct amp_ins_event {
    int32_t ref_pos;
    uint32_t read; /* row in the batch, plus the batch's read_base (amp_process_batch*) */
    int32_t q_from;
    int32_t q_to;
} amp_ins_event;

/*
 * Packed read batch.  Rows are the reads the reference's loop does not skip (mapped, with a
 * CIGAR: AmpliPy.py:902).  seq is 4-bit BAM codes "=ACMGRSVTWYHKDBN", high nibble first;
 * qual is Phred bytes, first byte 0xFF = QUAL '*'; l_seq 0 = SEQ '*'.  Offsets in `seq_off`
 * are in BASES and index both qual (bytes) and seq (nibbles); each must be a multiple of
 * AMP_SEQ_ALIGN.  cig is BAM's `len<<4 | op`.
 */
typedef struct amp_reads {
    int64_t n_reads;
    const int32_t *pos;       /* [n]   0-based leftmost coordinate (reference_start) */
    const uint16_t *flag;     /* [n]   SAM FLAG; bits 0x1 and 0x10 are read */
    const int32_t *tlen;      /* [n]   template_length */
    const uint32_t *lseq;     /* [n]   query_length */
    const uint64_t *cig_off;  /* [n+1] */
    const uint32_t *cig;      /* [cig_off[n]] */
    const uint64_t *seq_off;  /* [n+1] */
    const uint8_t *seq;       /* [seq_off[n]/2] */
    const uint8_t *qual;      /* [seq_off[n]] */
} amp_reads;

/*
 * Per-read results of trim_read (AmpliPy.py:426-687) and of the write filter's
 * reference_length (AmpliPy.py:910).  Read i's new CIGAR is written at
 * new_cig + cig_off[i] + 3*i, new_ncig[i] ops long (a trim adds at most 3 ops), so
 * new_cig needs cig_off[n] + 3*n entries.  Any pointer may be NULL to skip that output.
 */
typedef struct amp_trim_out {
    int32_t *new_pos;     /* [n] */
    uint32_t *new_ncig;   /* [n] */
    uint32_t *new_cig;    /* [cig_off[n] + 3n] */
    int32_t *ref_len;     /* [n] reference_length after trimming */
    uint8_t *trim_flags;  /* [n] AMP_TRIM_* bits */
    uint8_t *status;      /* [n] amp_read_status */
} amp_trim_out;

/* Same batch with every pointer in DEVICE memory and 32-bit offsets:
 * cig_off32[i] = cig_off[i], seq_off8[i] = seq_off[i] / 8.  seq and qual must be 8-byte aligned
 * and have 16 bytes of readable slack after the last read (the kernels use 8- and 16-byte
 * vector loads that may start in a read's last 8 bytes). */
typedef struct amp_dev_reads {
    int64_t n_reads;
    const int32_t *pos;
    const uint16_t *flag;
    const int32_t *tlen;
    const uint32_t *lseq;
    const uint32_t *cig_off32; /* [n+1] */
    const uint32_t *cig;
    const uint32_t *seq_off8;  /* [n+1] */
    const uint8_t *seq;
    const uint8_t *qual;
    int64_t n_cig;             /* cig_off32[n], known to the host */
    int64_t n_bases_padded;    /* padded bases of the rows: (seq_off8[n] - seq_off8[0]) * 8; only its mean per row is used (it picks
                                * the tile geometry of the fast kernel: a value that is too large costs speed, not correctness) */
} amp_dev_reads;

/* ---- library ------------------------------------------------------------------------ */
int amp_version(void);                        /* AMP_ABI_VERSION */
const char *amp_strerror(int rc);
const char *amp_last_error(const amp_ctx *ctx); /* text of the last failing HIP call */
const char *amp_read_status_exception(int status); /* "IndexError", "KeyError", ... */
int amp_device_count(void);

/* ---- primer tables: find_overlapping_primers, AmpliPy.py:174-209 (host, once per run) --
 * primers must be sorted ascending by (start, end) like AmpliPy.py:257.  Writes -1 for
 * None.  max_primer_len = max(end - start) (AmpliPy.py:876). */
int amp_find_overlapping_primers(int32_t ref_len, int32_t n_primers, const int32_t *starts,
                                 const int32_t *ends, int32_t primer_pos_offset,
                                 int32_t *min_primer_start, int32_t *max_primer_end,
                                 int32_t *max_primer_len);

/* ---- context -------------------------------------------------------------------------- */
/* Allocates the zeroed count table uint32[ref_len][AMP_NSYM] on `device` (AmpliPy.py:892). */
int amp_ctx_create(amp_ctx **out, int device, int32_t ref_len);
void amp_ctx_destroy(amp_ctx *ctx);
/* Optional: run all work of this ctx on the caller's HIP stream (hipStream_t). */
int amp_ctx_set_stream(amp_ctx *ctx, void *hip_stream);
/* Optional: use caller-owned DEVICE memory (uint32[ref_len*AMP_DEV_COLS], e.g. a torch
 * tensor) as the device table so torch.distributed can reduce it in place. Contents are kept. */
int amp_ctx_bind_counts(amp_ctx *ctx, void *dev_counts);

/* Tables consumed by trim_read (AmpliPy.py:450-452); host pointers, length ref_len. */
int amp_set_primers(amp_ctx *ctx, const int32_t *min_primer_start,
                    const int32_t *max_primer_end, int32_t max_primer_len);
/* min_quality / sliding_window_width of AmpliPy.py:907,915; do_trim = run_trim,
 * do_count = run_variants or run_consensus (AmpliPy.py:906, 914).
 * Any window >= 1 and any min_quality >= 0 give the reference's results; the FAST kernels (closed-form trims, one lane per
 * read) are built for windows of 1..8 bases and min_quality <= 128: a run outside of that takes the general tile kernel for
 * every read (about 1.5 x the time per batch, same results).  The third-generation kernel (variant 6) additionally needs
 * min_quality >= 1. */
int amp_set_params(amp_ctx *ctx, int32_t min_quality, int32_t window, int32_t do_trim,
                   int32_t do_count);

/* ---- the hot path: AmpliPy.py:896-915 for a whole batch --------------------------------
 * Host-pointer form: stages the batch to the device, runs the kernels, copies `out` back.
 * read_base is added to amp_ins_event.read so events of several batches stay distinct. */
int amp_process_batch(amp_ctx *ctx, const amp_reads *reads, uint64_t read_base,
                      const amp_trim_out *out);
/* Device-pointer form: inputs and outputs already resident in HBM; asynchronous on the
 * ctx stream (outputs are valid after amp_sync). */
int amp_process_batch_device(amp_ctx *ctx, const amp_dev_reads *reads, uint64_t read_base,
                             const amp_trim_out *dev_out);
int amp_sync(amp_ctx *ctx);

/* Time spent in the kernels of the last amp_process_batch* call, measured with HIP events
 * on the ctx stream: total, and the dominant (CIGAR-scan) kernel alone. */
int amp_last_kernel_ms(amp_ctx *ctx, float *total_ms, float *scan_ms);
/* split != 0: also time the first (dominant) kernel of every pass on its own; the extra event leaves the GPU idle for a
 * few microseconds behind that kernel, so it is off by default and scan_ms then repeats total_ms. */
int amp_set_timing(amp_ctx *ctx, int split);

/* ---- accumulated state ----------------------------------------------------------------- */
int amp_get_counts(amp_ctx *ctx, uint32_t *counts /* [ref_len][AMP_NSYM] host */);
int amp_add_counts(amp_ctx *ctx, const uint32_t *counts /* host, added element-wise */);
/* buf == NULL: *n = an upper bound of the events recorded so far (list slots in use; reads with long
 * CIGARs reserve a slice and may leave part of it unused).  buf != NULL (cap >= that bound): the events
 * are copied and *n = their exact number. */
int amp_get_ins_events(amp_ctx *ctx, int64_t *n, amp_ins_event *buf, int64_t cap);
/* The same, after which the event list is empty again (the per-position tally of amp_get_counts stays): for runs of many
 * batches, where the text of the events is taken batch by batch.  Call with buf == NULL first for the size, like above. */
int amp_drain_ins_events(amp_ctx *ctx, int64_t *n, amp_ins_event *buf, int64_t cap);
void *amp_counts_device_ptr(amp_ctx *ctx);
/* On-device aggregation of the insertion events recorded since the last amp_reset / drain (the dict
 * keys of AmpliPy.py:745-748, consumed at A:767-771): the device sorts the events by (ref_pos, allele) and run-length
 * encodes them.  One amp_ins_run per run of events with the same position and the same allele text: `count` events and one
 * representative (its text through amp_event_strings).  Two runs may carry the same allele (a 64-bit hash collision between
 * two alleles of one position and length, never seen): the consumer sums counts by (ref_pos, text); a run never mixes alleles.
 * reads = the device batch the events' read ids refer to, read_base as given to amp_process_batch* (reads == NULL: the batch
 * of the last amp_process_batch call, still staged on the device).  buf == NULL: *n_runs = an upper bound (list slots in
 * use); buf != NULL (cap >= that bound): the records, *n_runs = their number, sorted by (ref_pos, allele length).
 * drain != 0: the event list is empty afterwards (the per-position tally of amp_get_counts stays). */
typedef struct amp_ins_run {
    amp_ins_event first;
    uint32_t count;
    uint32_t reserved;
} amp_ins_run;
int amp_aggregate_ins_events(amp_ctx *ctx, const amp_dev_reads *reads, uint64_t read_base, int drain, int64_t *n_runs,
                             amp_ins_run *buf, int64_t cap);
/* Sum the device tables (counts + insertion tally) over the ranks of an RCCL communicator (ncclComm_t) onto rank `root`
 * (root < 0: all ranks).  comm == NULL is a no-op (single GPU). */
int amp_reduce(amp_ctx *ctx, void *rccl_comm, int root);
int amp_reset(amp_ctx *ctx); /* zero the count table and drop recorded events */
/* Number of reads with a non-zero amp_read_status since the last amp_reset. */
int amp_error_reads(amp_ctx *ctx, int64_t *n);
/* The reference's coordinate helpers for n CIGARs at once, computed on the device by the functions the kernels use:
 *   pos_on_query_out[i] = get_pos_on_query(cigar_i, ref_pos[i], ref_start[i])        (AmpliPy.py:389-412)
 *   pos_on_ref_out[i]   = get_pos_on_ref(cigar_i, query_pos[i], ref_start[i])        (AmpliPy.py:363-386)
 *   fixed_cig[cig_off[i] ..] / fixed_n[i] = fix_cigar(cigar_i)                       (AmpliPy.py:415-423)
 * cig_off[n + 1] / cig as in amp_reads (32-bit offsets, BAM words); status[i] = amp_read_status of get_pos_on_query in the
 * low nibble and of get_pos_on_ref in the high nibble (an op code >= 9 makes the reference's table look-up fail in the helper
 * that reaches it; fix_cigar consults no table and never fails).  Host pointers; synchronous. */
int amp_coordinate_helpers(amp_ctx *ctx, int64_t n, const uint32_t *cig_off, const uint32_t *cig, const int32_t *ref_start,
                           const int32_t *ref_pos, const int32_t *query_pos, int32_t *pos_on_query_out, int32_t *pos_on_ref_out,
                           uint32_t *fixed_cig, uint32_t *fixed_n, uint8_t *status);
/* Development aid: the 16 raw device counters ([0] events, [1] event bound, [2] error reads,
 * [3] deferred reads, [8..13] per-phase cycle sums when AMPLIHIP_PHASES has bit 0x100). */
int amp_debug_counters(amp_ctx *ctx, uint64_t *out16);
/* Development aid (AMPLIHIP_PHASES bit 0x100): per tile-kernel block {cycles, window rebases,
 * quality-scan chunks, counting chunks} of the last launch. */
int amp_debug_blocks(amp_ctx *ctx, uint32_t *out, int cap_blocks, int *n_blocks);
/* Pre-size the insertion-event buffer.  Without it every amp_process_batch* call first runs
 * a small bound kernel and synchronises to size the buffer; with it the call is fully
 * asynchronous and amp_get_ins_events reports AMP_EOVERFLOW if the reservation was short.
 * `cap` is per list shard (there are 8, and any of them may receive most of a batch's events); size it for twice the
 * events of a batch plus 64 slots per wave of the fast kernel (8 per CU) and of the many-op kernel (24 per CU): waves
 * reserve list slots 64 at a time and leave some unused (read-out drops them). */
int amp_reserve_events(amp_ctx *ctx, int64_t cap);
/* 0 (default) = chosen per batch between 4, 5 and 7 by its mean padded read length (up to 152: 4), the window (8: 5) and its mean number
 * of CIGAR ops (long reads with three ops a read and more: 7).  4 = the fast kernel (closed-form trim +
 * pileup of reads with one match op or one insertion / deletion of up to 152 bases, every byte loaded once) followed by the
 * general pass over the reads it hands over; 5 = its second generation (reads consumed from LDS staging buffers,
 * branch-free closed forms, reads of up to 304 bases); 6 = its third generation (reads of up to 160 bases sorted into class
 * lists per block, rows gathered by LDS-DMA, three passes from LDS: amp_fast6.hpp; opt-in); 7 = the second generation driven by
 * per-block lists of reads binned by length (tiles of one length class, two lanes per read of more than 144 bases, reads for the
 * general pass never in a tile: amp_fast7.hpp; for batches of mixed read lengths); 2 = the fused tile kernel over every read; 1 = one-lane-per-read
 * kernels and 3 = the tile kernel's work cut into three kernels -- 1 to 3 are kept for on-GPU A/B checks (all give
 * identical results).  Runs with window > 8 or min_quality > 128 use variant 2 whatever is set. */
int amp_set_kernel_variant(amp_ctx *ctx, int variant);
/* 1 when runs with the ctx's current parameters take a fast kernel (window 1..8, min_quality <= 128, variant 0 / 4 / 5 / 6 / 7), 0 when
 * every read takes the general tile kernel (same results, about 1.5 x the time); negative on a bad ctx.  Informational: lets a
 * caller that sweeps sliding-window widths know when it has left the fast path (AmpliPy.py:563 takes any width). */
int amp_fast_path_active(amp_ctx *ctx);
/* The kernel variant the last batch of the ctx took (what 0 = "chosen per batch" resolved to: 4, 5 or 7; 2 for runs outside of the
 * fast kernels' parameters; 0 before the first batch); negative on a bad ctx.  Informational. */
int amp_last_kernel_variant(amp_ctx *ctx);
/* Size the fast kernel's grid for 1 / divisor of the GPU's CUs (1, the default: one block per CU).  For callers that keep
 * several batches in flight on different streams (one ctx each): two passes side by side on half the chip each finish
 * sooner than one after the other on all of it -- a block then works twice as long, so its start-up, its flush and the idle
 * end of its last round of tiles weigh half as much (bench.py: 0.224 -> 0.212 ms per 2 M-read step with divisor 2, eight
 * steps in flight).  A pass that runs alone should keep the default.  Results do not depend on it. */
int amp_set_cu_share(amp_ctx *ctx, int divisor);

/* ---- calling: alleles_from_counts (AmpliPy.py:756-771) + the loop AmpliPy.py:917-952 --------
 * One device pass decides, for every reference position, everything that does not depend on
 * the TEXT of an insertion allele: total depth (all symbols, insertion events included,
 * :767), the six base symbols ranked like sorted(..., reverse=True) (:771), the consensus
 * symbol (:928-929) and the variant record (:933-951).  Frequencies are IEEE doubles
 * count/total compared with >=, exactly as Python computes them.
 * An insertion string can only change a decision when the position's insertion events could
 * out-rank the best base symbol or reach min_freq_variants; those positions come back with
 * AMP_CALL_INS_RELEVANT and the host finishes them from amp_get_ins_events (the strings live
 * in the reads: SEQ[q_from:q_to]).  full_ranking = 1 flags every position that has events. */
typedef struct amp_call_params {
    int32_t min_depth_consensus;
    int32_t min_depth_variants;
    double min_freq_consensus;
    double min_freq_variants;
    int32_t run_consensus;
    int32_t run_variants;
    int32_t full_ranking;
    int32_t reserved;
} amp_call_params;

#define AMP_CALL_VARIANT 1u      /* a VCF record is emitted (AmpliPy.py:940) */
#define AMP_CALL_GT_HAS_REF 2u   /* GT starts at 0 (AmpliPy.py:948-949) */
#define AMP_CALL_INS_RELEVANT 4u /* insertion alleles may change this position's outcome */

typedef struct amp_pos_call {
    uint32_t total_depth; /* :767 */
    uint32_t ref_count;   /* count of the reference symbol (:937); 0 when it is not one of A C G T N - */
    uint32_t order;       /* bits 3k..3k+2: column (A C G T N - = 0..5) of the k-th ranked base symbol;
                             bits 18..20: number of base symbols with a non-zero count */
    int8_t consensus_sym; /* -1 = unknown symbol, else column 0..5 of the consensus (:929) */
    uint8_t flags;        /* AMP_CALL_* */
    uint8_t alt_mask;     /* bit k: k-th ranked base symbol is an ALT (:938-939) */
    uint8_t pad;
} amp_pos_call;

/* One VCF record decided on the device (positions not flagged AMP_CALL_INS_RELEVANT). */
typedef struct amp_var_rec {
    int32_t pos;            /* 0-based; VCF POS = pos + 1 (AmpliPy.py:947) */
    uint32_t total_depth;   /* DP */
    uint32_t ref_count;     /* REF_DP */
    uint8_t n_alt;
    uint8_t gt_has_ref;     /* GT = 0/1/../n_alt when set, 1/../n_alt otherwise (AmpliPy.py:948-951) */
    uint8_t alt_col[6];     /* ALT symbols in ranked order: columns A C G T N - = 0..5 */
    uint32_t alt_count[6];  /* ALT_DP; ALT_FREQ = alt_count / total_depth as IEEE doubles */
} amp_var_rec;

/* ASCII reference sequence, length ref_len, exactly as read from the FASTA (REF is compared
 * un-upper-cased, AmpliPy.py:923). */
int amp_set_reference(amp_ctx *ctx, const uint8_t *ref_ascii);
int amp_call_positions(amp_ctx *ctx, const amp_call_params *params, amp_pos_call *out /* host [ref_len] */,
                       int64_t *n_relevant);
/* The same decisions packed for the host: consensus column per position (-1 unknown), the
 * variant records in ascending position, and the insertion-relevant positions (whose
 * consensus / record the host finishes).  Returns AMP_EOVERFLOW (with the needed counts) when
 * a capacity is too small; ref_len entries always suffice. */
int amp_call_compact(amp_ctx *ctx, const amp_call_params *params, int8_t *consensus /* [ref_len] */,
                     amp_var_rec *vars, int64_t vars_cap, int64_t *n_vars,
                     int32_t *relevant, int64_t relevant_cap, int64_t *n_relevant);
/* amp_call_compact without the last copy: the arrays stay in host memory owned by ctx (page-locked and written by the
 * kernel itself when the call was begun with amp_call_compact_begin, else an ordinary buffer filled by one copy) and are
 * valid until the next amp_call_* on the same ctx (or amp_ctx_destroy). */
typedef struct amp_call_view {
    const int8_t *consensus;       /* [ref_len] */
    const amp_var_rec *vars;       /* [n_vars] */
    const int32_t *relevant;       /* [n_relevant] */
    int64_t n_vars, n_relevant;
} amp_call_view;
int amp_call_compact_view(amp_ctx *ctx, const amp_call_params *params, amp_call_view *out);
/* Enqueue the work of amp_call_compact_view on the ctx stream without waiting (right behind amp_process_batch_device,
 * say): a following amp_call_compact_view with the same parameters only waits for it and hands the views out.  Lets a
 * caller with several contexts in flight keep the calling kernels of one step in front of the next step's reads.
 * begin must come BEHIND the last update of the table: amp_process_batch[_device], amp_reset, amp_add_counts, amp_reduce
 * and amp_ctx_bind_counts all cancel a begun call (the following view then computes afresh), so on several GPUs the order
 * is process, amp_reduce, begin, view.  From begin onward the memory behind the previous view is being overwritten. */
int amp_call_compact_begin(amp_ctx *ctx, const amp_call_params *params);
/* Text of insertion events from a DEVICE-resident batch: text[off[e] .. off[e+1]) receives
 * SEQ[q_from:q_to] of event e (off[e+1]-off[e] must equal q_to-q_from). ev/off/text are host.
 * reads == NULL: the batch of the last amp_process_batch call (its device copy stays in the ctx until the next one;
 * AMP_ESTATE when there was none) -- a caller that feeds host batches gets the allele text of a batch's events
 * without gathering the bases on the host (AmpliPy.py:736-738 builds each string from the read it is looking at). */
int amp_event_strings(amp_ctx *ctx, const amp_dev_reads *reads, uint64_t read_base, int64_t n_events,
                      const amp_ins_event *events, const uint64_t *off, uint8_t *text);

#ifdef __cplusplus
}
#endif
#endif /* AMPLIHIP_H */
