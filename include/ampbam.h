/* ampbam.h -- host-side BAM/BGZF decode into the packed read batch of amplihip.h, and re-encode
 * of trimmed records.  Plain C ABI, no GPU, no torch types.
 *
 * Replaces, for BAM files, what AmpliPy v0.0.2 gets from pysam (a dependency that is not part of
 * /root/reference: pysam 0.17.0 / htslib 1.13, requirements.txt:1):
 *   pysam.AlignmentFile(fn, 'rb') + iteration           AmpliPy.py:296-324, :896
 *   the skip rule `is_unmapped or cigartuples is None`  AmpliPy.py:902
 *   pysam.AlignmentFile(fn, 'wb', header=...) / write   AmpliPy.py:326-356, :911
 * The format is restated from the SAM/BAM specification (SAMv1 section 4): BGZF = concatenated gzip
 * members of <= 64 KiB with a `BC` extra subfield holding the block size; a BAM record is
 * block_size, 32 bytes of fixed fields, read name, uint32 CIGAR (len<<4|op), 4-bit bases (high
 * nibble first), qualities (0xFF... = absent), auxiliary fields.
 *
 * Every function returns 0 or a negative ampbam_rc; nothing throws across the ABI.  A handle is
 * not thread-safe; the library uses its own worker threads inside a call.
 *
 * Memory: buffers of 4 MB and more (inflated images, the packed batch, the writer's blocks) are not returned to the system
 * when a handle is closed but kept for the next handle of the process, up to AMPBAM_POOL_MB megabytes in all (environment,
 * default 4096; 0 = keep nothing).  The arrays of a decoded batch are what a GPU runtime copies from, and un-mapping pages it
 * has mapped for DMA stalls the process's next GPU call by tens of milliseconds; a file walked piece by
 * piece also finds its next piece's buffers already faulted in.  Other environment switches: AMPBAM_ZLIB=1 (never use
 * libdeflate), AMPBAM_ZLIB_INFLATE=1 / AMPBAM_ZLIB_CRC=1 (zlib's inflate / crc32 instead of the codec's own, which are tried first
 * and checked by the block CRC), AMPBAM_HUGEPAGES=1 (MADV_HUGEPAGE on those buffers).
 */
#ifndef AMPBAM_H
#define AMPBAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ampbam_file ampbam_file;       /* a BAM file, fully inflated in host memory */
typedef struct ampbam_writer ampbam_writer;

enum ampbam_rc {
    AMPBAM_OK = 0,
    AMPBAM_EINVAL = -1,   /* bad argument */
    AMPBAM_EIO = -2,      /* open / read / write failed */
    AMPBAM_EFORMAT = -3,  /* not BGZF / not BAM / truncated / CRC mismatch */
    AMPBAM_ENOMEM = -4
};

int ampbam_version(void);
/* CRC-32 (gzip polynomial) of a buffer, as the codec computes it for the BGZF blocks it reads and writes (carry-less
 * multiplication folding on x86 with PCLMULQDQ, zlib's crc32 otherwise: the two must agree, and the tests compare them). */
uint32_t ampbam_crc32(const void *data, int64_t n_bytes);
/* The codec's own DEFLATE decoder (amplipy_amd/csrc/amp_inflate.hpp) on one raw stream whose inflated size is known, as it is
 * run on every BGZF block before zlib gets a block it refuses: 0 when the stream ends with its final block after exactly
 * n_out bytes, AMPBAM_EFORMAT otherwise (never reads or writes outside the two buffers).  Exposed for the tests. */
int ampbam_inflate_raw(const void *in, int64_t n_in, void *out, int64_t n_out);
const char *ampbam_strerror(int rc);

/* ---- reading ---------------------------------------------------------------------------- */
/* Reads `path`, inflates all BGZF blocks with `n_threads` workers (<= 0: one per available CPU,
 * at most 16), checks each block's CRC32, parses the header and indexes the records. */
int ampbam_open(const char *path, int n_threads, ampbam_file **out);
/* Part `part` of `n_parts` of the file (0 <= part < n_parts): a rank of a multi-GPU run inflates only its share, and a
 * single process can walk a file piece by piece with bounded memory (AmpliPy.py:896 streams).  The cut points are
 * compressed-byte offsets rounded up to BGZF block starts -- equal shares of bases for a coordinate-sorted BAM of similar
 * reads -- and a part owns the records that START inside its blocks (the blocks its last record runs into are inflated too).
 * A BAM file does not say where records start inside a block: the first record of a part > 0 is the first offset from which
 * a chain of 64 plausible records runs.  To make that exact, compare neighbours with ampbam_part_range: part k + 1 must
 * start where part k ended (a mismatch: fall back to ampbam_open).  Record numbers of such a file count from the part's
 * first record; header text and references are those of the whole file.  n_parts == 1 is ampbam_open. */
int ampbam_open_range(const char *path, int n_threads, int part, int n_parts, ampbam_file **out);
/* The same when the caller KNOWS where the part's first record starts -- the inflated offset at which the part before it ended
 * (ampbam_part_range of that part): nothing is guessed, which is how one process walks a file piece by piece (only a rank's
 * first piece needs the heuristic above).  first_hint == UINT64_MAX: ampbam_open_range.  A hint at or behind the part's last
 * block means no record starts in the part: an empty part whose range is (hint, hint).  AMPBAM_EINVAL for a hint in front of
 * the part's first block, AMPBAM_EFORMAT when no chain of records runs from the hint to the part's end. */
int ampbam_open_range_at(const char *path, int n_threads, int part, int n_parts, uint64_t first_hint, ampbam_file **out);
/* Offsets in the file's INFLATED stream of the part's first record and of the byte behind its last record (equal for a
 * part without records; 0 / 0 ... for a file opened with ampbam_open: header end / stream end are not tracked there). */
int ampbam_part_range(const ampbam_file *f, uint64_t *first, uint64_t *end);
void ampbam_close(ampbam_file *f);
const char *ampbam_last_error(const ampbam_file *f);

int64_t ampbam_n_records(const ampbam_file *f);
/* SAM header text (not NUL-terminated; *len bytes) and the reference dictionary. */
int ampbam_header_text(const ampbam_file *f, const char **text, int64_t *len);
int32_t ampbam_n_refs(const ampbam_file *f);
int ampbam_ref(const ampbam_file *f, int32_t i, const char **name, int32_t *length);

/* Records [first, first+count) decoded into the packed structure-of-arrays batch that
 * amp_process_batch takes (same field meaning as struct amp_reads in amplihip.h).  Rows are the
 * records the reference's loop does not skip (mapped, at least one CIGAR op; AmpliPy.py:902), in
 * file order; src_index[i] is the record number of row i.  The arrays belong to `f` and stay valid
 * until the next ampbam_decode / ampbam_close on it. */
typedef struct ampbam_batch {
    int64_t n_reads;
    const int32_t *pos;        /* 0-based leftmost reference position */
    const uint16_t *flag;
    const int32_t *tlen;
    const uint32_t *lseq;
    const uint64_t *cig_off;   /* [n_reads+1] into cig */
    const uint32_t *cig;       /* len<<4|op */
    const uint64_t *seq_off;   /* [n_reads+1], in bases, every read starts on a multiple of 8 */
    const uint8_t *seq;        /* 4-bit codes, high nibble first; seq_off/2 bytes into it */
    const uint8_t *qual;       /* one byte per base at seq_off; first byte 0xFF = no qualities */
    const int64_t *src_index;  /* [n_reads] record numbers */
    int64_t n_cig, n_bases;    /* totals: cig_off[n_reads], seq_off[n_reads]; seq and qual carry 16 spare bytes */
    int64_t n_skipped;         /* records of the range left out by the skip rule */
} ampbam_batch;
int ampbam_decode(ampbam_file *f, int64_t first, int64_t count, ampbam_batch *out);

/* ---- writing ---------------------------------------------------------------------------- */
/* A new BAM file with the given SAM header text and the reference dictionary of `like`
 * (AmpliPy.py:343-346 copies the input's references).  level = zlib level 0..9 (-1: default). */
int ampbam_writer_open(const char *path, const char *header_text, int64_t header_len, const ampbam_file *like,
                       int level, int n_threads, ampbam_writer **out);
/* Appends rows of a decoded batch whose keep[i] != 0, re-encoded with a new position and CIGAR:
 * row i of the batch is record src_index[i] of `src`; its new CIGAR is new_ncig[i] words at
 * new_cig + new_cig_off[i].  block_size, bin (reg2bin of [pos, end)), n_cigar_op and pos change,
 * everything else (name, flag, mapq, mate fields, bases, qualities, aux) is copied. */
int ampbam_write_rows(ampbam_writer *w, const ampbam_file *src, int64_t n_rows, const int64_t *src_index,
                      const uint8_t *keep, const int32_t *new_pos, const uint32_t *new_ncig,
                      const uint64_t *new_cig_off, const uint32_t *new_cig);
/* The new file starts with BGZF blocks that hold the header and nothing else: their size in the file.  The files written by the
 * ranks of a multi-GPU run are joined into the ONE file AmpliPy.py writes (A:326-356, A:911) by copying part 0 without its
 * EOF block and the later parts without their header blocks and EOF blocks, and ending with one EOF block (BGZF members
 * concatenate). */
int64_t ampbam_writer_header_bytes(const ampbam_writer *w);
/* Appends the rows of a packed batch (the SoA of ampbam_decode / include/amplihip.h: 4-bit bases, one quality byte per base,
 * BAM CIGAR words) as NEW records: name "r<name_base + row>", reference 0, MAPQ 60, mate on the read's own position, no aux
 * fields.  For files made from synthetic or re-packed reads (the benchmarks' input files; tests); AmpliPy itself only ever
 * re-writes records it has read (ampbam_write_rows, A:911).  seq_off in bases (even), as in the batch. */
int ampbam_write_batch(ampbam_writer *w, int64_t n, const int32_t *pos, const uint16_t *flag, const int32_t *tlen, const uint32_t *lseq,
                       const uint64_t *cig_off, const uint32_t *cig, const uint64_t *seq_off, const uint8_t *seq, const uint8_t *qual,
                       uint64_t name_base);
/* Flushes, writes the BGZF end-of-file block, closes the file and frees the writer. */
int ampbam_writer_close(ampbam_writer *w);

#ifdef __cplusplus
}
#endif
#endif
