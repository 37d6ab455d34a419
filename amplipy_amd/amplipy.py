"""Host mirror of AmpliPy's command-line / run_amplipy surface on top of the HIP engine.

Same sub-commands, flags, defaults, log lines and error behaviour as AmpliPy.py v0.0.2
(parse_args AmpliPy.py:113-171, run_amplipy :774-963, __main__ :966-1025); the per-read loop
(:896-915) is replaced by batches handed to libamplihip.so, calling (:917-952) by
``amplipy_amd.calling``, and pysam I/O by ``amplipy_amd.bamio``.  There is no CPU execution
path: without a GPU the engine raises.
"""
from __future__ import annotations

import argparse
import gzip
import os
import sys
from datetime import datetime
from os.path import isfile

import numpy as np

from . import AMPLIPY_VERSION, abi, bamio, calling, lib, parallel
from .batch import ReadBatch
from .insertions import EventStore

VERSION = AMPLIPY_VERSION
PROGRESS_NUM_READS = 50000          # AmpliPy.py:19
BATCH_READS = 1 << 18              # reads per device batch on the Python-codec path (one Rec object each)
NATIVE_BATCH_READS = 1 << 18       # records per device batch on the libampbam path (the writer overlaps the next batch)

DEFAULTS = dict(min_depth_consensus=10, min_depth_variants=1, min_freq_consensus=0, min_freq_variants=0.03,
                min_length=30, min_quality=20, primer_pos_offset=0, sliding_window_width=4, unknown_symbol="N")


def print_log(s="", end="\n"):
    print("[%s] %s" % (datetime.now().strftime("%Y-%m-%d %H:%M:%S"), s), end=end, file=sys.stderr)
    sys.stderr.flush()


def error(s=None):
    print_log("ERROR" if s is None else "ERROR: %s" % s)
    sys.exit(1)


# ---- loaders (AmpliPy.py:212-258) --------------------------------------------------------------
def load_ref_genome(reference_fn):
    if not isfile(reference_fn):
        error("File not found: %s" % reference_fn)
    with open(reference_fn) as f:
        lines = f.read().strip().splitlines()
    if len(lines) < 2 or not lines[0].startswith(">"):
        error("Invalid FASTA file: %s" % reference_fn)
    ref_id = lines[0][1:].split()[0].strip()
    seq = "".join(lines[1:])
    if ">" in seq:
        error("Multiple sequences in FASTA file: %s" % reference_fn)
    return ref_id, seq


def load_primers(primer_fn):
    if not isfile(primer_fn):
        error("File not found: %s" % primer_fn)
    with open(primer_fn) as f:
        lines = f.read().strip().splitlines()
    primers = []
    for l in lines:
        parts = l.split("\t")
        try:
            if len(parts) != 4:
                raise ValueError
            primers.append((int(parts[1]), int(parts[2])))
        except ValueError:
            error("Invalid primer BED line: %s" % l)
    if not primers:
        raise NameError("name 'header' is not defined")      # what the reference does on an empty BED (:255)
    primers.sort()
    return primers


# ---- output openers (AmpliPy.py:261-360) --------------------------------------------------------
def _reads_mode(fn, write):
    low = fn.lower()
    if low.endswith(".sam"):
        return "w" if write else "r"
    if low.endswith(".bam"):
        return "wb" if write else "rb"
    error("Invalid read mapping extension (should be .sam or .bam): %s" % fn)


def open_alignment_files(input_fn, output_fn):
    if input_fn is None:
        error("Input alignment file is None")
    if input_fn.lower() == "stdin":
        reader = bamio.AlignmentReader("-", "r")
    elif not isfile(input_fn):
        error("File not found: %s" % input_fn)
    else:
        reader = bamio.AlignmentReader(input_fn, _reads_mode(input_fn, False))
    writer = None
    if output_fn is not None:
        hdr = reader.header.with_amplipy_pg(VERSION, " ".join(sys.argv))
        if output_fn.lower() == "stdout":
            writer = bamio.AlignmentWriter("-", "w", hdr)
        elif isfile(output_fn):
            error("File already exists: %s" % output_fn)
        else:
            writer = bamio.AlignmentWriter(output_fn, _reads_mode(output_fn, True), hdr)
    return reader, writer


NATIVE_PART_BYTES = 4 << 20        # compressed bytes of a piece of the input BAM (pieces are inflated one ahead of the GPU); on the
                                   # 11.6 MB / 1.5 M-read file of the bench: 16 MB (one piece) aio 10.5 M reads/s, 4 MB 12.4, 1 MB 12.4


def native_parts(input_fn, rank=0, world=1, part_bytes=None):
    """How a BAM file is cut for this rank: (n_parts, k_lo, k_hi) -- the file has n_parts pieces of about ``part_bytes``
    compressed bytes (ampbam_open_range: cut at BGZF block starts, i.e. by base count for a sorted BAM), of which the rank
    takes the contiguous run [k_lo, k_hi).  Every rank gets the same number of pieces (n_parts is a multiple of world)."""
    part_bytes = part_bytes or int(os.environ.get("AMPLIPY_PART_BYTES", NATIVE_PART_BYTES))
    size = os.path.getsize(input_fn)
    per_rank = max(1, -(-size // (part_bytes * world)))
    return per_rank * world, per_rank * rank, per_rank * (rank + 1)


def open_native_bam(input_fn, output_fn, rank=0, world=1):
    """(NativeInput, BamWriter or None) when the native codec can serve this run: BAM file in, and BAM file
    (or nothing) out.  None otherwise (SAM text, stdin / stdout): the Python codec handles those.
    Same checks and messages as open_alignment_files."""
    if input_fn is None or input_fn.lower() == "stdin" or not isfile(input_fn) or _reads_mode(input_fn, False) != "rb":
        return None
    if output_fn is not None and (output_fn.lower() == "stdout" or isfile(output_fn) or _reads_mode(output_fn, True) != "wb"):
        return None
    if os.environ.get("AMPLIPY_PYTHON_BAM"):
        return None
    src = NativeInput(input_fn, rank, world)
    writer = None
    if output_fn is not None:
        from . import bam_native
        first = src.first_part()
        hdr = bamio.Header(first.header_text, first.references).with_amplipy_pg(VERSION, " ".join(sys.argv))
        # zlib's default level like htslib; AMPLIPY_BAM_LEVEL=1 trades file size for speed
        writer = bam_native.BamWriter(output_fn, hdr.text, first, level=int(os.environ.get("AMPLIPY_BAM_LEVEL", "-1")))
    return src, writer


class NativeInput:
    """The rank's share of a BAM file as a sequence of pieces (bam_native.BamFile of ampbam_open_range), each inflated and
    indexed on a helper thread while the piece before it is on the GPU (AmpliPy.py:896 streams its input; here at most two
    pieces are in memory).  Pieces must meet: each starts where the one before ended (checked; ranks check their seams with
    each other through ``seam``)."""

    def __init__(self, path, rank=0, world=1):
        self.path = path
        self.n_parts, self.k_lo, self.k_hi = native_parts(path, rank, world)
        self._first = None
        self._ahead = None          # (thread, box) of the piece being opened

    def _open(self, k, first_hint=None):
        from . import bam_native
        return bam_native.BamFile(self.path, part=k, n_parts=self.n_parts, first_hint=first_hint)

    def first_part(self):
        if self._first is None:
            self._first = self._open(self.k_lo)
        return self._first

    def _start(self, k, first_hint):
        import threading
        box = {}

        def run():
            try:
                box["file"] = self._open(k, first_hint)
            except Exception as e:           # surfaced by the consumer
                box["error"] = e
        t = threading.Thread(target=run, daemon=True)
        t.start()
        self._ahead = (t, box)

    def close(self):
        """Lets go of what an abandoned walk still holds: the piece that was being opened ahead, the first piece if it was never
        yielded."""
        if self._ahead is not None:
            t, box = self._ahead
            t.join()
            if box.get("file") is not None:
                box["file"].close()
            self._ahead = None
        if self._first is not None:
            self._first.close()
            self._first = None

    def __iter__(self):
        """Yields the pieces in order; the caller closes each when it is done with it.  seam = (first, end) of the whole
        share is available afterwards."""
        prev_end = None
        self.seam = [None, None]
        cur = self.first_part()
        self._first = None
        for k in range(self.k_lo, self.k_hi):
            # the piece behind this one starts where this one ends: it is told so, and only the rank's FIRST piece (whose
            # predecessor another rank reads) is found by the codec's chain-of-plausible-records search
            a, b = cur.part_range()
            if k + 1 < self.k_hi:
                exact = cur.n_records > 0 or k > self.k_lo or self.k_lo == 0      # (a guessed piece without records does not know where it ends)
                self._start(k + 1, b if exact else None)
            if cur.n_records:
                if prev_end is not None and a != prev_end:
                    raise bam_native_error("%s: piece %d of %d starts at inflated offset %d, the piece before it ended at %d"
                                           % (self.path, k, self.n_parts, a, prev_end))
                if self.seam[0] is None:
                    self.seam[0] = a
                self.seam[1] = prev_end = b
            yield cur
            if k + 1 < self.k_hi:
                t, box = self._ahead
                t.join()
                self._ahead = None
                if "error" in box:
                    raise box["error"]
                cur = box["file"]


def bam_native_error(msg):
    from . import bam_native
    return bam_native.AmpBamError(msg)


class VcfWriter:
    """Text VCF with the header AmpliPy builds through pysam (AmpliPy.py:271-281)."""

    def __init__(self, fn, ref_id):
        if fn.lower() == "stdout":
            self.f = sys.stdout
        elif isfile(fn):
            error("File already exists: %s" % fn)
        elif fn.lower().endswith(".vcf"):
            self.f = open(fn, "w")
        elif fn.lower().endswith(".vcf.gz"):
            self.f = gzip.open(fn, "wt")
        elif fn.lower().endswith(".bcf"):
            error("BCF output is not supported by this build (use .vcf or .vcf.gz): %s" % fn)
        else:
            error("Invalid variants extension (should be .vcf, .vcf.gz, or .bcf): %s" % fn)
        self.ref_id = ref_id
        w = self.f.write
        w("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n")
        w("##AmpliPyVersion=%s\n##source=%s\n##contig=<ID=%s>\n" % (VERSION, " ".join(sys.argv), ref_id))
        w("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        w("##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total Depth\">\n")
        w("##INFO=<ID=REF_DP,Number=1,Type=Integer,Description=\"Depth of reference base\">\n")
        w("##INFO=<ID=ALT_DP,Number=1,Type=String,Description=\"Depth of alternate base\">\n")
        w("##INFO=<ID=REF_FREQ,Number=1,Type=Float,Description=\"Frequency of reference base\">\n")
        w("##INFO=<ID=ALT_FREQ,Number=1,Type=String,Description=\"Frequency of alternate base\">\n")
        w("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample\n")

    def line(self, r):
        return "%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%d;REF_DP=%d;ALT_DP=%s;REF_FREQ=%g;ALT_FREQ=%s\tGT\t%s\n" % (
            self.ref_id, r.pos + 1, r.ref, ",".join(r.alts), r.DP, r.REF_DP, r.ALT_DP, r.REF_FREQ, r.ALT_FREQ, "/".join(map(str, r.GT)))

    def write(self, r):
        self.f.write(self.line(r))

    def write_all(self, records):
        self.f.write("".join([self.line(r) for r in records]))

    def close(self):
        if self.f is not sys.stdout:
            self.f.close()


def _raise_for_status(status):
    exc = abi.READ_STATUS_EXC[int(status)]
    raise exc("read rejected by the engine: %s (the reference raises %s here)" % (abi.READ_STATUS_NAMES[int(status)], exc.__name__))


def _store_events(eng, ins_store, read_base):
    """A batch's insertion alleles (A:730-748) into the store: the device sorts the batch's events by (position, allele) and
    run-length encodes them (amp_aggregate_ins_events), the text of one representative per allele is gathered on
    the device from the copy of the batch that eng.process() left there (A:736-738), and the event list starts over."""
    runs = eng.aggregate_events(read_base=read_base, drain=True)
    if runs.size == 0:
        return
    rows = np.zeros(runs.size, abi.INS_EVENT_DTYPE)
    for f in ("ref_pos", "q_from", "q_to"):
        rows[f] = runs[f]
    rows["read"] = (runs["read"].astype(np.int64) - (read_base & 0xFFFFFFFF)) & 0xFFFFFFFF     # read ids are 32-bit, relative to read_base modulo 2^32
    length, blob = eng.event_text(rows, 0)
    ins_store.add_text(runs["ref_pos"], length, blob, runs["count"])


def run_amplipy(untrimmed_reads_fn=None, primer_fn=None, reference_fn=None, trimmed_reads_fn=None, variants_fn=None,
                consensus_fn=None, primer_pos_offset=None, min_length=None, min_quality=None, sliding_window_width=None,
                min_freq_consensus=None, min_freq_variants=None, min_depth_consensus=None, min_depth_variants=None,
                unknown_symbol=None, include_no_primer=None, run_trim=False, run_variants=False, run_consensus=False,
                device=None):
    """The reference's run_amplipy (AmpliPy.py:774-963) on the MI355X engine.

    One process drives one GPU.  Under ``torchrun`` (WORLD_SIZE > 1, or AMPLIPY_FORCE_DIST=1 for a one-rank
    rehearsal) the job is range-partitioned: rank r takes the r-th contiguous run of BAM records (coordinate
    order), every rank's count table is summed with ONE all-reduce over RCCL (parallel.allreduce_table), the
    insertion alleles of the flagged positions are exchanged, and rank 0 writes the VCF / consensus.  A trimmed
    BAM is written per rank (``<name>.part<rank>.bam``; the parts concatenate to the single-GPU file's records).
    """
    dist, rank, world = parallel.init_from_env()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if dist is not None else 0
    # argument checks and banner: AmpliPy.py:836-866
    if primer_pos_offset is not None and primer_pos_offset < 0:
        error("Primer position offset must be non-negative: %s" % primer_pos_offset)
    if min_length is not None and min_length < 1:
        error("Minimum length must be >= 1: %s" % min_length)
    if min_quality is not None and min_quality < 0:
        error("Minimum quality must be non-negative: %s" % min_quality)
    if sliding_window_width is not None and sliding_window_width < 1:
        error("Sliding window width must be >= 1: %s" % sliding_window_width)
    for v in (min_freq_consensus, min_freq_variants):
        if v is not None and (v < 0 or v > 1):
            error("Minimum frequency must be between 0 and 1: %s" % v)
    for v in (min_depth_consensus, min_depth_variants):
        if v is not None and v < 0:
            error("Minimum depth must be positive: %s" % v)
    if unknown_symbol is not None and len(unknown_symbol) != 1:
        error("Unknown symbol must be exactly 1 character: %s" % unknown_symbol)
    if not (run_trim or run_variants or run_consensus):
        error("Not running any of the AmpliPy operations")
    mode = "Trim" if run_trim and not (run_variants or run_consensus) else \
        "Variants" if run_variants and not (run_trim or run_consensus) else \
        "Consensus" if run_consensus and not (run_trim or run_variants) else "All-In-One"
    print_log("Executing AmpliPy %s (v%s)" % (mode, VERSION))

    ref_id = ref_seq = None
    if reference_fn is not None:
        print_log("Loading reference genome: %s" % reference_fn)
        ref_id, ref_seq = load_ref_genome(reference_fn)
    G = len(ref_seq)
    eng = lib.Engine(G, device=device)
    if os.environ.get("AMPLIPY_DEV") == "1" and os.environ.get("AMPLIPY_KERNEL_VARIANT"):      # (A/B checks of the tests: all variants give the same files)
        eng.set_kernel_variant(int(os.environ["AMPLIPY_KERNEL_VARIANT"]))
    table = None
    final_trimmed_fn = None
    if dist is not None:
        import torch
        torch.cuda.set_device(device)
        table = torch.zeros(G * 7, dtype=torch.int32, device="cuda:%d" % device)   # counts + insertion tally
        eng.bind_counts(table.data_ptr())
        if run_trim and trimmed_reads_fn is not None and world > 1:
            # every rank writes the trimmed reads of its share to a file of its own; rank 0 joins them into the ONE file the
            # caller asked for once every rank is done (BGZF members concatenate: bam_native.stitch_bam_parts)
            if rank == 0 and trimmed_reads_fn.lower() != "stdout" and isfile(trimmed_reads_fn):
                error("File already exists: %s" % trimmed_reads_fn)
            final_trimmed_fn = trimmed_reads_fn
            root, ext = os.path.splitext(trimmed_reads_fn)
            trimmed_reads_fn = "%s.part%d%s" % (root, rank, ext)
    if primer_fn is not None:
        print_log("Loading primers: %s" % primer_fn)
        primers = load_primers(primer_fn)
        print_log("Precalculating overlapping primers...")
        mn, mx, mpl = lib.find_overlapping_primers(G, primers, primer_pos_offset)
        eng.set_primers(mn, mx, mpl)
    # Opening the files can fail on ONE rank of a multi-rank run (a missing share, an output that exists): the other ranks must
    # not be left waiting in the collective, so with several ranks the failure is carried to the exchange in front of it
    native = reader = writer = vcf = None
    rank_error = None
    try:
        if run_trim:
            print_log("Input untrimmed SAM/BAM: %s" % untrimmed_reads_fn)
            print_log("Output trimmed SAM/BAM: %s" % trimmed_reads_fn)
            native = open_native_bam(untrimmed_reads_fn, trimmed_reads_fn, rank, world)
            if native is None:
                reader, writer = open_alignment_files(untrimmed_reads_fn, trimmed_reads_fn)
        else:
            print_log("Input trimmed SAM/BAM: %s" % trimmed_reads_fn)
            native = open_native_bam(trimmed_reads_fn, None, rank, world)
            if native is None:
                reader, writer = open_alignment_files(trimmed_reads_fn, None)
        if variants_fn is not None and rank == 0:
            print_log("Output variants VCF: %s" % variants_fn)
            vcf = VcfWriter(variants_fn, ref_id)
    except (Exception, SystemExit) as e:
        if dist is None:
            raise
        rank_error = e if isinstance(e, Exception) else RuntimeError("could not open the run's files (exit status %s)" % (e.code,))
        native = None
    do_count = run_variants or run_consensus
    eng.set_params(min_quality if min_quality is not None else 20,
                   sliding_window_width if sliding_window_width is not None else 4, run_trim, do_count)

    print_log("Processing reads...")
    writer_pending = False      # the writer thread of the native BAM path is still running (joined behind the calls)
    ins_store = EventStore()             # insertion events with their allele text (each batch's bases are at hand only now)
    pending = []
    s_i = None
    read_base = 0

    def flush():
        nonlocal read_base
        if not pending:
            return
        batch = ReadBatch.from_segments([r.to_segment() for r in pending])
        res = eng.process(batch, read_base=read_base)
        bad = np.nonzero(res.status)[0]
        good_until = int(bad[0]) if len(bad) else len(pending)
        if run_trim and writer is not None:
            for k, r in enumerate(pending[:good_until]):
                fl = int(res.trim_flags[k])
                if int(res.ref_len[k]) >= min_length and ((fl & 3) or include_no_primer):      # AmpliPy.py:910
                    writer.write(r, pos=int(res.new_pos[k]), cigar=res.cigar_ops(k))
        if len(bad):                      # the reference dies on the first such read with an uncaught exception,
            _raise_for_status(res.status[bad[0]])      # having written every read in front of it (A:907-911)
        if do_count:
            # this batch's events only: the list is drained batch by batch (read ids are 32-bit and relative to
            # read_base modulo 2^32, which a batch never spans)
            _store_events(eng, ins_store, read_base)
        read_base += batch.n
        del pending[:]

    n_seen = 0                           # records this rank has gone through (all of them when there is one rank)
    n_bases = 0                          # ... and their bases (the measure the shares of a multi-rank run should be equal in: SURVEY 8e)
    seam = [None, None]
    if rank_error is not None:
        pass                              # (nothing was opened: straight to the exchange)
    elif native is not None:
        # BAM in (and BAM or nothing out): libampbam decodes records straight into packed batches and
        # re-encodes the kept ones; no per-read Python object exists on this path.  The file is walked piece by piece:
        # piece k + 1 is inflated and indexed on a helper thread while piece k is decoded, trimmed and counted, and a
        # writer thread re-encodes and deflates the rows of piece k - 1 (every stage is a C call that releases the GIL;
        # rows stay in order)
        src, nwriter = native
        wq = werr = wthread = None
        if run_trim and nwriter is not None:
            import queue
            import threading
            wq = queue.Queue(maxsize=3); werr = []

            def _writer():
                while True:
                    job = wq.get()
                    if job is None:
                        return
                    try:
                        if job[0] == "close":
                            job[1].close()
                        elif not werr:
                            nwriter.write_rows(*job[1:])
                    except Exception as e:       # surfaced by the main thread
                        werr.append(e)
            wthread = threading.Thread(target=_writer, daemon=True); wthread.start()
        loop_done = False
        try:
            for piece in src:
                for first in range(0, piece.n_records, NATIVE_BATCH_READS):
                    count = min(NATIVE_BATCH_READS, piece.n_records - first)
                    batch, _ = piece.decode(first, count)
                    for k_ in range(n_seen + (-n_seen) % PROGRESS_NUM_READS, n_seen + count, PROGRESS_NUM_READS):
                        if k_:
                            print_log("Processed %d reads..." % k_)
                    n_seen += count
                    s_i = n_seen - 1
                    if batch.n == 0:
                        continue
                    n_bases += int(batch.lseq.sum(dtype=np.int64))
                    res = eng.process(batch, read_base=read_base)
                    bad = np.nonzero(res.status)[0]
                    if wq is not None:
                        if werr:
                            raise werr[0]
                        keep = (res.ref_len >= min_length) & (((res.trim_flags & 3) != 0) | bool(include_no_primer))   # AmpliPy.py:910
                        if len(bad):
                            keep[int(bad[0]):] = False          # the reads in front of the failing one are still written (A:907-911)
                        slot_off = batch.cig_off[:-1] + np.uint64(3) * np.arange(batch.n, dtype=np.uint64)
                        # src_index is a view of the decoder's buffers, which the next decode overwrites
                        wq.put(("rows", piece, batch.src_index.copy(), keep, res.new_pos, res.new_ncig, slot_off, res.new_cig))
                    if len(bad):
                        _raise_for_status(res.status[bad[0]])
                    if do_count:
                        _store_events(eng, ins_store, read_base)
                    read_base += batch.n
                if wq is not None:
                    wq.put(("close", piece))        # (the writer copies the unchanged parts of a record from the piece's image)
                else:
                    piece.close()
            loop_done = True
        except Exception as e:                      # with several ranks the others must not be left waiting in the collective
            if dist is None:
                raise
            rank_error = e
        finally:
            if not loop_done:
                src.close()                 # (the piece opened ahead of the one that failed)
            if wq is not None:
                wq.put(None)
                # One process: the writer thread goes on re-encoding and deflating the last rows under the calls and the VCF
                # text below (32 ms of Python for 12,000 records) and is joined behind them.  Several ranks need to know
                # whether it failed before the collective.
                if dist is not None or not loop_done:
                    wthread.join()
                else:
                    writer_pending = True
        if not writer_pending:
            if werr and rank_error is None:
                if dist is None:
                    raise werr[0]
                rank_error = werr[0]
            if nwriter is not None and rank_error is None:
                nwriter.close()
        seam = getattr(src, "seam", [None, None])
    else:
        seam = [None, None]
        try:
            for s_i, rec in enumerate(reader):
                if s_i % PROGRESS_NUM_READS == 0 and s_i != 0:
                    print_log("Processed %d reads..." % s_i)
                if (rec.flag & 4) or rec.cigar is None:            # AmpliPy.py:902
                    continue
                if world > 1 and s_i % world != rank:              # text input has no record index: deal the reads out
                    continue
                pending.append(rec)
                if len(pending) >= BATCH_READS:
                    flush()
            flush()
            if writer is not None:
                writer.close()
            reader.close()
        except Exception as e:
            if dist is None:
                raise
            rank_error = e

    if dist is not None:
        # before the one collective of the run: did every rank get through its share (a rank that raised must not leave the
        # others waiting in the all-reduce), and do the shares of neighbouring ranks meet (every share of a BAM file starts
        # where the one before it ended: what makes the split of ampbam_open_range exact)
        trouble = parallel.exchange_notes(dist, world, seam, rank_error)
        if trouble:
            eng.close()
            parallel.finish(dist)
            if rank_error is not None:
                raise rank_error
            raise RuntimeError(trouble)
        shares = parallel.gather_objects(dist, rank, world, (n_seen, n_bases))
        if rank == 0 and native is not None:
            # the file is cut by compressed bytes (ampbam_open_range): what that gave every rank, in records and in bases
            tot = max(sum(b_ for _, b_ in shares), 1)
            print_log("Shares of the %d ranks: records %s; bases %s (%s %% of the job)" % (world, [r_ for r_, _ in shares], [b_ for _, b_ in shares],
                                                                                        ", ".join("%.1f" % (100.0 * b_ / tot) for _, b_ in shares)))
        if final_trimmed_fn is not None and native is not None and native[1] is not None:
            # the ranks' files (all closed by now: the writer threads were joined above) become the one trimmed BAM
            parts = parallel.gather_objects(dist, rank, world, (native[1].path, native[1].header_bytes))
            if rank == 0:
                from . import bam_native
                bam_native.stitch_bam_parts(final_trimmed_fn, parts)
                for path, _ in parts:
                    os.remove(path)
                print_log("Trimmed reads of %d ranks joined: %s" % (world, final_trimmed_fn))

    try:
      if do_count:
        cp = calling.call_params(min_depth_consensus if min_depth_consensus is not None else 0,
                                 min_freq_consensus if min_freq_consensus is not None else 0,
                                 min_depth_variants if min_depth_variants is not None else 0,
                                 min_freq_variants if min_freq_variants is not None else 0,
                                 run_consensus, run_variants)
        eng.set_reference(ref_seq)
        if dist is not None:
            eng.sync()
            parallel.allreduce_table(dist, table)      # the ONE collective of the run: every rank now holds the job's table

        def ins_tallies(positions):
            triples = ins_store.counted_pairs(positions)
            if dist is not None:                       # all ranks flag the same positions (same table): symmetric exchange
                triples = parallel.allgather_relevant_events(dist, world, triples)
            return calling.tallies_from_runs(triples, positions)
        res = calling.call(eng, ref_seq, cp, ins_tallies)
        if rank != 0:
            run_variants = run_consensus = False       # rank 0 writes the outputs
        if run_variants:
            vcf.f.write(res.vcf_text(vcf.ref_id))        # (= vcf.write(r) for r in res.records)
            vcf.close()
        if run_consensus:
            f = gzip.open(consensus_fn, "wt") if consensus_fn.lower().endswith(".gz") else open(consensus_fn, "w")
            f.write(">sample\n%s\n" % res.consensus_string(unknown_symbol))
            f.close()
    finally:
        if writer_pending:
            wthread.join()
            if nwriter is not None and sys.exc_info()[0] is not None:
                try:                        # (the calls failed: the trimmed BAM is still ended properly)
                    nwriter.close()
                except Exception:
                    pass
    if writer_pending:
        if werr:
            raise werr[0]
        if nwriter is not None:
            nwriter.close()
    eng.close()
    parallel.finish(dist)
    if s_i is None:
        if dist is None or world == 1:
            raise NameError("name 's_i' is not defined")       # the reference's behaviour on an empty input (:963)
        s_i = -1                                               # (a rank whose share is empty: more ranks than pieces of the file)
    print_log("Finished Processing %d reads" % s_i)


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        argv.append("-h")
    D = DEFAULTS
    fmt = argparse.ArgumentDefaultsHelpFormatter
    parser = argparse.ArgumentParser(description="amplipy_amd: AmpliPy's toolkit surface on MI355X", formatter_class=fmt)
    sub = parser.add_subparsers(dest="command")

    def trim_args(p):
        p.add_argument("-p", "--primer", required=True, type=str, help="Primer File (BED)")
        p.add_argument("-r", "--reference", required=True, type=str, help="Reference Genome (FASTA)")

    t = sub.add_parser("trim", formatter_class=fmt)
    t.add_argument("-i", "--input", required=False, type=str, default="stdin", help="Untrimmed Reads (SAM/BAM)")
    trim_args(t)
    t.add_argument("-o", "--output", required=False, type=str, default="stdout", help="Trimmed Reads (SAM/BAM)")
    t.add_argument("-x", "--primer_pos_offset", required=False, type=int, default=D["primer_pos_offset"], help="Primer position offset")
    t.add_argument("-ml", "--min_length", required=False, type=int, default=D["min_length"], help="Minimum length of read to retain after trimming")
    t.add_argument("-mq", "--min_quality", required=False, type=int, default=D["min_quality"], help="Minimum quality threshold")
    t.add_argument("-s", "--sliding_window_width", required=False, type=int, default=D["sliding_window_width"], help="Width of sliding window")
    t.add_argument("-e", "--include_no_primer", action="store_true", help="Include reads with no primers")

    v = sub.add_parser("variants", formatter_class=fmt)
    v.add_argument("-i", "--input", required=False, type=str, default="stdin", help="Trimmed Reads (SAM/BAM)")
    v.add_argument("-r", "--reference", required=True, type=str, help="Reference Genome (FASTA)")
    v.add_argument("-o", "--output", required=False, type=str, default="stdout", help="Variant Calls (VCF)")
    v.add_argument("-mq", "--min_quality", required=False, type=int, default=D["min_quality"], help="Minimum quality threshold")
    v.add_argument("-mf", "--min_freq", required=False, type=float, default=D["min_freq_variants"], help="Minimum frequency threshold (0-1) to call variant")
    v.add_argument("-md", "--min_depth", required=False, type=int, default=D["min_depth_variants"], help="Minimum depth to call variant")

    c = sub.add_parser("consensus", formatter_class=fmt)
    c.add_argument("-i", "--input", required=False, type=str, default="stdin", help="Trimmed Reads (SAM/BAM)")
    c.add_argument("-r", "--reference", required=True, type=str, help="Reference Genome (FASTA)")
    c.add_argument("-o", "--output", required=False, type=str, default="stdout", help="Consensus Sequence (FASTA)")
    c.add_argument("-mq", "--min_quality", required=False, type=int, default=D["min_quality"], help="Minimum quality threshold")
    c.add_argument("-mf", "--min_freq", required=False, type=float, default=D["min_freq_consensus"], help="Minimum frequency threshold (0-1) to call consensus")
    c.add_argument("-md", "--min_depth", required=False, type=int, default=D["min_depth_consensus"], help="Minimum depth to call consensus")
    c.add_argument("-n", "--unknown_symbol", required=False, type=str, default=D["unknown_symbol"], help="Character to print in regions with less than minimum coverage")

    a = sub.add_parser("aio", formatter_class=fmt)
    a.add_argument("-i", "--input", required=False, type=str, default="stdin", help="Untrimmed Reads (SAM/BAM)")
    trim_args(a)
    a.add_argument("-ot", "--output_trimmed_reads", required=True, type=str, help="Trimmed Reads (SAM/BAM)")
    a.add_argument("-ov", "--output_variants", required=True, type=str, help="Variant Calls (VCF)")
    a.add_argument("-oc", "--output_consensus", required=True, type=str, help="Consensus Sequence (FASTA)")
    a.add_argument("-x", "--primer_pos_offset", required=False, type=int, default=D["primer_pos_offset"], help="Primer position offset")
    a.add_argument("-ml", "--min_length", required=False, type=int, default=D["min_length"], help="Minimum length of read to retain after trimming")
    a.add_argument("-mq", "--min_quality", required=False, type=int, default=D["min_quality"], help="Minimum quality threshold")
    a.add_argument("-s", "--sliding_window_width", required=False, type=int, default=D["sliding_window_width"], help="Width of sliding window")
    a.add_argument("-mfc", "--min_freq_consensus", required=False, type=float, default=D["min_freq_consensus"], help="Minimum frequency threshold (0-1) to call consensus")
    a.add_argument("-mfv", "--min_freq_variants", required=False, type=float, default=D["min_freq_variants"], help="Minimum frequency threshold (0-1) to call variant")
    a.add_argument("-mdc", "--min_depth_consensus", required=False, type=int, default=D["min_depth_consensus"], help="Minimum depth to call consensus")
    a.add_argument("-mdv", "--min_depth_variants", required=False, type=int, default=D["min_depth_variants"], help="Minimum depth to call variant")
    a.add_argument("-n", "--unknown_symbol", required=False, type=str, default=D["unknown_symbol"], help="Character to print in regions with less than minimum coverage")
    a.add_argument("-e", "--include_no_primer", action="store_true", help="Include reads with no primers")
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.command == "trim":
        run_amplipy(untrimmed_reads_fn=args.input, primer_fn=args.primer, reference_fn=args.reference,
                    trimmed_reads_fn=args.output, primer_pos_offset=args.primer_pos_offset, min_length=args.min_length,
                    min_quality=args.min_quality, sliding_window_width=args.sliding_window_width,
                    include_no_primer=args.include_no_primer, run_trim=True)
    elif args.command == "variants":
        run_amplipy(trimmed_reads_fn=args.input, reference_fn=args.reference, variants_fn=args.output,
                    min_quality=args.min_quality, min_freq_variants=args.min_freq, min_depth_variants=args.min_depth,
                    run_variants=True)
    elif args.command == "consensus":
        run_amplipy(trimmed_reads_fn=args.input, reference_fn=args.reference, consensus_fn=args.output,
                    min_quality=args.min_quality, min_freq_consensus=args.min_freq, min_depth_consensus=args.min_depth,
                    unknown_symbol=args.unknown_symbol, run_consensus=True)
    elif args.command == "aio":
        run_amplipy(untrimmed_reads_fn=args.input, primer_fn=args.primer, reference_fn=args.reference,
                    trimmed_reads_fn=args.output_trimmed_reads, variants_fn=args.output_variants,
                    consensus_fn=args.output_consensus, primer_pos_offset=args.primer_pos_offset,
                    min_length=args.min_length, min_quality=args.min_quality,
                    sliding_window_width=args.sliding_window_width, min_freq_consensus=args.min_freq_consensus,
                    min_freq_variants=args.min_freq_variants, min_depth_consensus=args.min_depth_consensus,
                    min_depth_variants=args.min_depth_variants, unknown_symbol=args.unknown_symbol,
                    include_no_primer=args.include_no_primer, run_trim=True, run_variants=True, run_consensus=True)


if __name__ == "__main__":
    main()
