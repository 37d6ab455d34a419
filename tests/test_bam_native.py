"""libampbam (include/ampbam.h): the native BAM decode / re-encode against the Python codec of
bamio.py on the same files.  No GPU needed."""
import ctypes as C
import gzip
import os
import re

import numpy as np
import pytest

from amplipy_amd import bam_native, bamio, synth
from amplipy_amd.batch import ReadBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_bam(path, n=700, seed=11, with_oddities=True):
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    segs = synth.make_mixed_segments(g, amps, n, seed=seed)
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:SYN_REF\tLN:%d\n@PG\tID:sim\tPN:sim\tVN:1\tCL:sim x\n" % g.size
    hdr = bamio.Header(text, [("SYN_REF", int(g.size))])
    recs = []
    for i, s in enumerate(segs):
        recs.append(bamio.Rec("r%d" % i, s.flag, 0, s.reference_start, 60, s.cigartuples, 0, s.reference_start, s.template_length,
                              s.query_sequence, bytes(s.query_qualities), aux_sam=["NM:i:%d" % (i % 7), "XZ:Z:abc"]))
    if with_oddities:
        recs.insert(5, bamio.Rec("unmapped", 4, -1, -1, 0, None, -1, -1, 0, "ACGTN", bytes([30] * 5)))          # skipped (A:902)
        recs.insert(9, bamio.Rec("nocigar", 0, 0, 100, 0, None, -1, -1, 0, "ACGT", bytes([30] * 4)))               # skipped: no CIGAR
        recs.insert(12, bamio.Rec("noqual", 0, 0, 200, 60, [(0, 7)], -1, -1, 0, "ACGTACG", None))                   # QUAL '*'
        recs.insert(15, bamio.Rec("odd", 16, 0, 300, 60, [(4, 2), (0, 9)], -1, -1, 0, "ACGTACGTANN"[:11], bytes(range(11))))
    w = bamio.AlignmentWriter(path, "wb", hdr)
    for r in recs:
        w.write(r)
    w.close()
    return hdr, recs


def test_header_declares_exactly_the_exports():
    L = bam_native.load()
    decl = re.findall(r"\b(ampbam_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "ampbam.h")).read())
    assert sorted(set(decl)) == sorted(bam_native.EXPORTS)
    assert L.ampbam_version() == 1


def test_decode_matches_python_codec(tmp_path):
    bam = str(tmp_path / "a.bam")
    hdr, recs = _make_bam(bam)
    f = bam_native.BamFile(bam, threads=3)
    assert f.n_records == len(recs) and f.header_text == hdr.text and f.references == hdr.refs
    kept = [(i, r) for i, r in enumerate(recs) if not (r.flag & 4) and r.cigar is not None]
    want = ReadBatch.from_segments([r.to_segment() for _, r in kept])
    got, skipped = f.decode(0, f.n_records, copy=True)
    assert skipped == len(recs) - len(kept) == 2
    assert got.src_index.tolist() == [i for i, _ in kept]
    for name in ("pos", "flag", "tlen", "lseq", "cig_off", "cig", "seq_off", "seq", "qual"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    # record ranges: any split gives the same rows
    a, _ = f.decode(0, 10, copy=True)
    b, _ = f.decode(10, f.n_records - 10, copy=True)
    assert np.array_equal(np.concatenate([a.pos, b.pos]), want.pos) and a.n + b.n == want.n
    assert np.array_equal(np.concatenate([a.cig, b.cig]), want.cig)
    assert np.array_equal(np.concatenate([a.src_index, b.src_index]), got.src_index)
    f.close()


def test_reencode_matches_python_writer(tmp_path):
    bam = str(tmp_path / "a.bam"); out_n = str(tmp_path / "n.bam"); out_p = str(tmp_path / "p.bam")
    hdr, recs = _make_bam(bam)
    f = bam_native.BamFile(bam, threads=2)
    batch, _ = f.decode(0, f.n_records, copy=True)
    rng = np.random.default_rng(3)
    n = batch.n
    keep = (rng.random(n) < 0.8).astype(np.uint8)
    new_pos = batch.pos + rng.integers(0, 30, n).astype(np.int32)
    # new CIGARs in slots of old length + 3, like the device writes them
    slot_off = (batch.cig_off[:-1] + 3 * np.arange(n, dtype=np.uint64)).astype(np.uint64)
    new_cig = np.zeros(int(batch.cig.size) + 3 * n, np.uint32)
    new_ncig = np.zeros(n, np.uint32)
    cigars = []
    for i in range(n):
        k = int(rng.integers(1, 4))
        ops = [(int(rng.choice([0, 1, 2, 4, 7, 8])), int(rng.integers(1, 40))) for _ in range(k)]
        cigars.append(ops); new_ncig[i] = k
        new_cig[int(slot_off[i]):int(slot_off[i]) + k] = [(ln << 4) | op for op, ln in ops]
    hdr2 = hdr.with_amplipy_pg("0.0.2", "amplipy trim")
    w = bam_native.BamWriter(out_n, hdr2.text, f, threads=2)
    half = n // 2                                                   # two calls, like two batches
    w.write_rows(None, batch.src_index[:half], keep[:half], new_pos[:half], new_ncig[:half], slot_off[:half], new_cig)
    w.write_rows(None, batch.src_index[half:], keep[half:], new_pos[half:], new_ncig[half:], slot_off[half:], new_cig)
    w.close()
    pw = bamio.AlignmentWriter(out_p, "wb", hdr2)
    rd = bamio.AlignmentReader(bam, "rb")
    all_recs = list(rd)
    for i in range(n):
        if keep[i]:
            pw.write(all_recs[int(batch.src_index[i])], pos=int(new_pos[i]), cigar=cigars[i])
    pw.close()
    raw_n = gzip.decompress(open(out_n, "rb").read()); raw_p = gzip.decompress(open(out_p, "rb").read())
    assert raw_n == raw_p                                           # byte-identical BAM streams
    assert open(out_n, "rb").read().endswith(bamio.BGZF_EOF)
    back = list(bamio.AlignmentReader(out_n, "rb"))
    assert len(back) == int(keep.sum())
    f.close()


def test_bad_input_is_refused(tmp_path):
    p = str(tmp_path / "x.bam")
    open(p, "wb").write(b"not a bam at all")
    with pytest.raises(bam_native.AmpBamError):
        bam_native.BamFile(p)
    bam = str(tmp_path / "a.bam")
    _make_bam(bam, n=50, with_oddities=False)
    raw = bytearray(open(bam, "rb").read())
    raw[40] ^= 0x55                                                  # corrupt the first block's payload
    open(p, "wb").write(bytes(raw))
    with pytest.raises(bam_native.AmpBamError):
        bam_native.BamFile(p)
    with pytest.raises(bam_native.AmpBamError):
        bam_native.BamFile(str(tmp_path / "missing.bam"))


def test_open_range_parts_tile_the_file(tmp_path):
    """ampbam_open_range (a rank inflates only its share; a file can be read piece by piece): for several
    numbers of parts, every part's first record starts where the part before it ended (the check that makes the heuristic
    record-start search exact), the last part ends at the end of the inflated stream, and the decoded rows of all parts,
    concatenated, are the rows of the whole file.  Two writers: the Python codec (records straddle BGZF blocks) and
    libampbam's own."""
    from amplipy_amd import bam_native, synth
    from tools.e2e_legs import write_bam
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    hb = synth.make_amplicon_batch(g, amps, 12000, seed=5)
    p1 = str(tmp_path / "a.bam")
    write_bam(p1, hb, int(g.size))
    whole = bam_native.BamFile(p1)
    wb, _ = whole.decode(0, whole.n_records, copy=True)
    # a second file written by libampbam itself from the first
    p2 = str(tmp_path / "b.bam")
    w = bam_native.BamWriter(p2, whole.header_text, whole, level=1)
    off = wb.cig_off[:-1].copy()
    w.write_rows(None, wb.src_index, np.ones(wb.n, np.uint8), wb.pos, np.diff(wb.cig_off.astype(np.int64)).astype(np.uint32), off, wb.cig)
    w.close()
    for path in (p1, p2):
        for n_parts in (2, 3, 5, 8, 64):
            parts = [bam_native.BamFile(path, part=k, n_parts=n_parts) for k in range(n_parts)]
            rng = [p.part_range() for p in parts]
            nonempty = [k for k in range(n_parts) if parts[k].n_records]
            assert sum(p.n_records for p in parts) == whole.n_records, (path, n_parts)
            for a, b in zip(nonempty[:-1], nonempty[1:]):
                assert rng[a][1] == rng[b][0], (path, n_parts, a, b)
            assert all(p.header_text == whole.header_text and p.references == whole.references for p in parts)
            rows = [p.decode(0, p.n_records, copy=True)[0] for p in parts if p.n_records]
            assert np.array_equal(np.concatenate([r.pos for r in rows]), wb.pos)
            assert np.array_equal(np.concatenate([r.cig for r in rows]), wb.cig)
            assert np.array_equal(np.concatenate([r.qual for r in rows]), wb.qual)
            assert np.array_equal(np.concatenate([r.seq for r in rows]), wb.seq)
            for p in parts:
                p.close()
    whole.close()


def test_buffers_handed_from_file_to_file(tmp_path):
    """The codec keeps its big buffers (inflated image, packed batch, writer blocks: 4 MB and more) for the next file instead
    of returning them to the system.  A file decoded into buffers that still hold another file's bytes must give exactly
    what a fresh process gives (the Python codec's rows), padding included, in any order of opening and closing, and a
    stream written through reused blocks must inflate to the Python writer's."""
    big = str(tmp_path / "big.bam"); small = str(tmp_path / "small.bam")
    hb, rb = _make_bam(big, n=26000, seed=5, with_oddities=False)
    hs, rs = _make_bam(small, n=23000, seed=6, with_oddities=True)

    def expected(recs):
        return ReadBatch.from_segments([r.to_segment() for r in recs if not (r.flag & 4) and r.cigar is not None])

    def same(got, want):
        for name in ("pos", "flag", "tlen", "lseq", "cig_off", "cig", "seq_off", "seq", "qual"):
            assert np.array_equal(getattr(got, name), getattr(want, name)), name
    wb, ws = expected(rb), expected(rs)
    assert wb.qual.size >= (4 << 20) and ws.qual.size >= (4 << 20)          # these buffers do go through the pool
    for path, want in ((big, wb), (small, ws), (big, wb), (small, ws)):
        f = bam_native.BamFile(path, threads=4)
        got, _ = f.decode(0, f.n_records)
        same(got, want)
        src = got.src_index.copy()                    # (the arrays of a decode are views: the next decode overwrites them)
        # ... and a second, shorter decode into the same arrays (stale rows behind it must not show)
        part, _ = f.decode(100, 5000)
        r0 = int(np.searchsorted(src, 100)); r1 = int(np.searchsorted(src, 5100))
        assert part.n == r1 - r0 and np.array_equal(part.pos, want.pos[r0:r1]) and np.array_equal(part.lseq, want.lseq[r0:r1])
        lo, hi = int(want.seq_off[r0]), int(want.seq_off[r1])
        assert np.array_equal(part.qual[:hi - lo], want.qual[lo:hi]) and np.array_equal(part.seq[:(hi - lo) // 2], want.seq[lo // 2:hi // 2])
        f.close()
    # two files open at once, closed in the other order
    fa = bam_native.BamFile(big, threads=2); fb = bam_native.BamFile(small, threads=2)
    ga, _ = fa.decode(0, fa.n_records); gb, _ = fb.decode(0, fb.n_records)
    same(ga, wb); same(gb, ws)
    # a stream written through reused blocks
    out_n = str(tmp_path / "n.bam"); out_p = str(tmp_path / "p.bam")
    n = gb.n
    keep = np.ones(n, np.uint8); keep[::7] = 0
    w = bam_native.BamWriter(out_n, hs.text, fb, threads=4)
    w.write_rows(None, gb.src_index, keep, gb.pos, np.diff(gb.cig_off.astype(np.int64)).astype(np.uint32), gb.cig_off[:-1], gb.cig)
    w.close()
    pw = bamio.AlignmentWriter(out_p, "wb", hs)
    for i, r in enumerate(r for r in rs if not (r.flag & 4) and r.cigar is not None):
        if keep[i]:
            pw.write(r)
    pw.close()
    assert gzip.decompress(open(out_n, "rb").read()) == gzip.decompress(open(out_p, "rb").read())
    fa.close(); fb.close()


def test_block_crc_equals_zlibs():
    """ampbam_crc32 (PCLMULQDQ folding where the CPU has it) against zlib.crc32 on every length from 0 to 300 and on random
    lengths up to a BGZF block and beyond, unaligned starts included; AMPBAM_ZLIB_CRC=1 selects zlib's for comparison runs."""
    import zlib
    L = bam_native.load()
    rng = np.random.default_rng(9)
    buf = rng.integers(0, 256, 200000, dtype=np.uint8)
    base = buf.ctypes.data
    lens = list(range(0, 301)) + [int(x) for x in rng.integers(301, 70000, 300)] + [65280, 65536, 199999]
    for n in lens:
        off = int(rng.integers(0, 17)) if n + 17 < buf.size else 0
        got = int(L.ampbam_crc32(C.c_void_p(base + off), C.c_int64(n)))
        assert got == (zlib.crc32(buf[off:off + n].tobytes()) & 0xFFFFFFFF if n else 0), (n, off)


def test_own_inflate_against_zlib():
    """amp_inflate.hpp (the codec's DEFLATE decoder, tried on every BGZF block before zlib) against zlib on raw streams of every
    compression level and strategy -- stored, fixed-Huffman and dynamic blocks, several blocks per stream, runs (distance 1),
    long and overlapping matches, incompressible bytes, BAM-like records -- and on damaged input: a truncated stream, flipped
    bits, random bytes and a wrong expected size must be refused or give bytes that the caller's CRC check then rejects, and
    never touch memory outside the two buffers (the guard bytes around the output stay intact)."""
    import zlib
    L = bam_native.load()
    rng = np.random.default_rng(17)

    def payloads():
        yield b""
        yield b"a"
        yield b"abc" * 5000
        yield bytes(60000)
        yield rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()
        yield rng.integers(0, 4, 65280, dtype=np.uint8).tobytes()
        yield bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 40000)) + bytes(rng.choice([37, 37, 37, 25, 11, 2], 25000).astype(np.uint8))
        rec = bytes(rng.integers(0, 256, 36, dtype=np.uint8)) + b"read_name_%05d\0" + bytes([0x60, 0x09, 0, 0]) + bytes(rng.integers(0, 256, 75, dtype=np.uint8)) + bytes([37] * 150)
        yield b"".join(rec.replace(b"%05d", b"%05d" % i) for i in range(220))
        for n in (1, 2, 7, 8, 9, 255, 256, 257, 258, 259, 300, 4095, 32768, 32769, 65535):
            yield bytes(rng.integers(0, 3, n, dtype=np.uint8))

    def inflate(raw, n_out, pad=64):
        out = np.full(n_out + 2 * pad, 0xA5, np.uint8)
        src = np.frombuffer(raw, np.uint8).copy() if raw else np.zeros(1, np.uint8)
        rc = L.ampbam_inflate_raw(C.c_void_p(src.ctypes.data), C.c_int64(len(raw)), C.c_void_p(out.ctypes.data + pad), C.c_int64(n_out))
        assert (out[:pad] == 0xA5).all() and (out[pad + n_out:] == 0xA5).all(), "wrote outside the output buffer"
        return rc, out[pad:pad + n_out].tobytes()

    n_ok = 0
    for data in payloads():
        for level in (0, 1, 3, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                half = len(data) // 2
                raw = co.compress(data[:half]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(data[half:]) + co.flush()
                rc, got = inflate(raw, len(data))
                assert rc == 0 and got == data, (len(data), level, strategy)
                n_ok += 1
                if len(data) > 300 and level == 6 and strategy == zlib.Z_DEFAULT_STRATEGY:
                    assert inflate(raw, len(data) - 1)[0] != 0 and inflate(raw, len(data) + 1)[0] != 0          # wrong ISIZE
                    assert inflate(raw[:len(raw) // 2], len(data))[0] != 0                                       # truncated
                    for _ in range(40):                                                                          # flipped bits
                        bad = bytearray(raw); k = int(rng.integers(0, len(bad))); bad[k] ^= 1 << int(rng.integers(0, 8))
                        rc2, got2 = inflate(bytes(bad), len(data))
                        assert rc2 != 0 or got2 == data or zlib.crc32(got2) != zlib.crc32(data)
    assert n_ok > 500
    for _ in range(300):                                                                                         # random bytes
        raw = rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8).tobytes()
        inflate(raw, int(rng.integers(0, 70000)))


def test_record_index_ignores_decoy_records(tmp_path):
    """The record index is built by several threads, each of which has to GUESS where a record starts in its stretch of the
    inflated stream (64 plausible records in a row) before the stretches are stitched along the true chain.  Here long reads
    carry, in their quality bytes, runs of 70 perfectly plausible fake records: threads that start on one must be overruled.
    Every number of threads gives the Python codec's rows."""
    import struct
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    segs = synth.make_mixed_segments(g, amps, 16000, seed=3)
    hdr = bamio.Header("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:SYN_REF\tLN:%d\n" % g.size, [("SYN_REF", int(g.size))])
    fake = struct.pack("<iiiBBHHHIiii", 34, 0, 5, 2, 60, 4680, 0, 0, 0, -1, -1, 0) + b"A\0"
    assert len(fake) == 38
    decoy_q = (fake * 70 + bytes([30]) * 40)
    decoy_q = bytes([30]) * 3 + decoy_q                      # (a first byte of 0xFF would mean "no qualities")
    recs = []
    for i, s in enumerate(segs):
        recs.append(bamio.Rec("r%d" % i, s.flag, 0, s.reference_start, 60, s.cigartuples, 0, s.reference_start, s.template_length,
                              s.query_sequence, bytes(s.query_qualities)))
        if i % 150 == 75:
            L = len(decoy_q)
            recs.append(bamio.Rec("decoy%d" % i, 0, 0, 100 + i % 1000, 60, [(0, L)], -1, -1, 0, "ACGT" * (L // 4) + "A" * (L % 4), decoy_q))
    bam = str(tmp_path / "d.bam")
    w = bamio.AlignmentWriter(bam, "wb", hdr)
    for r in recs:
        w.write(r)
    w.close()
    want = ReadBatch.from_segments([r.to_segment() for r in recs])
    for threads in (1, 2, 5, 8, 16):
        f = bam_native.BamFile(bam, threads=threads)
        assert f.n_records == len(recs), threads
        got, skipped = f.decode(0, f.n_records)
        assert skipped == 0
        for name in ("pos", "flag", "lseq", "cig_off", "cig", "seq_off", "seq", "qual"):
            assert np.array_equal(getattr(got, name), getattr(want, name)), (threads, name)
        f.close()
    # ... and as parts of a file
    first_prev = None
    total = 0
    for k in range(5):
        f = bam_native.BamFile(bam, threads=4, part=k, n_parts=5)
        a, b = f.part_range()
        assert first_prev is None or a == first_prev
        first_prev = b
        total += f.n_records
        f.close()
    assert total == len(recs)


def test_piece_walk_does_not_guess_record_starts(tmp_path, monkeypatch):
    """A decoy that ENDS where its host record ends -- a trailing B:C tag whose bytes are a run of plausible records -- rejoins the
    true chain, so a piece that guessed its first record inside the tag would look right to every local check and only fail
    against its neighbour (one record too many).  One process walking a file never guesses: every piece after the first is
    opened at the offset where the piece before it ended (ampbam_open_range_at).  The walk over many small pieces gives the
    whole file's rows; the hinted open of every part equals the part's range; and a header longer than the first parts'
    blocks leaves those parts empty instead of failing."""
    import struct
    from amplipy_amd import amplipy
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    segs = synth.make_mixed_segments(g, amps, 6000, seed=11)
    hdr = bamio.Header("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:SYN_REF\tLN:%d\n" % g.size, [("SYN_REF", int(g.size))])
    fake = struct.pack("<iiiBBHHHIiii", 34, 0, 5, 2, 60, 4680, 0, 0, 0, -1, -1, 0) + b"A\0"
    tag = b"zzBC" + struct.pack("<I", 80 * len(fake)) + fake * 80            # aux array of 80 plausible records: the record ends with it
    recs = []
    for i, s in enumerate(segs):
        r = bamio.Rec("r%d" % i, s.flag, 0, s.reference_start, 60, s.cigartuples, 0, s.reference_start, s.template_length,
                      s.query_sequence, bytes(s.query_qualities))
        if i % 97 == 13:
            r.aux_bam = tag
        recs.append(r)
    bam = str(tmp_path / "t.bam")
    w = bamio.AlignmentWriter(bam, "wb", hdr)
    for r in recs:
        w.write(r)
    w.close()
    whole = bam_native.BamFile(bam)
    assert whole.n_records == len(recs)
    wb, _ = whole.decode(0, whole.n_records, copy=True)
    whole.close()
    # the walk of run_amplipy's input: many pieces, each told where it starts
    monkeypatch.setenv("AMPLIPY_PART_BYTES", "20000")
    src = amplipy.NativeInput(bam, 0, 1)
    assert src.n_parts > 20
    rows, n = [], 0
    for piece in src:
        n += piece.n_records
        if piece.n_records:
            rows.append(piece.decode(0, piece.n_records, copy=True)[0])
        piece.close()
    assert n == len(recs)
    assert np.array_equal(np.concatenate([r.pos for r in rows]), wb.pos) and np.array_equal(np.concatenate([r.qual for r in rows]), wb.qual)
    # hinted opens agree with the chain of part ranges; a wrong hint is refused or runs into garbage, never accepted silently as another range
    n_parts = 9
    end = None
    total = 0
    for k in range(n_parts):
        f = bam_native.BamFile(bam, part=k, n_parts=n_parts, first_hint=end)
        a, b = f.part_range()
        assert end is None or a == end
        end = b
        total += f.n_records
        f.close()
    assert total == len(recs)
    # a header that is longer than the first parts
    big = bamio.Header("@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:c%06d\tLN:1000\n" % i for i in range(60000)) ,
                       [("c%06d" % i, 1000) for i in range(60000)])
    bam2 = str(tmp_path / "h.bam")
    w = bamio.AlignmentWriter(bam2, "wb", big)
    for r in recs[:500]:
        w.write(r)
    w.close()
    for n_parts in (1, 4, 16):
        end = None; total = 0
        for k in range(n_parts):
            f = bam_native.BamFile(bam2, part=k, n_parts=n_parts, first_hint=end)
            end = f.part_range()[1]
            total += f.n_records
            f.close()
        assert total == 500, n_parts


def test_write_batch_round_trip(tmp_path):
    """ampbam_write_batch (files made from packed rows: the benchmarks' inputs): odd and even read lengths, indel CIGARs, a read
    without qualities; libampbam's own reader and the Python codec read back what was packed."""
    from tools.e2e_legs import write_bam
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    segs = synth.make_mixed_segments(g, amps, 3000, seed=21)
    segs[5].query_qualities = None
    hb = ReadBatch.from_segments(segs)
    seed = str(tmp_path / "seed.bam")
    write_bam(seed, synth.make_amplicon_batch(g, amps, 8, seed=1), int(g.size))
    like = bam_native.BamFile(seed)
    out = str(tmp_path / "o.bam")
    w = bam_native.BamWriter(out, like.header_text, like, level=1)
    w.write_batch(hb, name_base=1000)
    w.close(); like.close()
    f = bam_native.BamFile(out)
    assert f.n_records == hb.n
    b, skipped = f.decode(0, f.n_records, copy=True)
    assert skipped == 0
    for name in ("pos", "flag", "tlen", "lseq", "cig_off", "cig", "seq_off", "seq", "qual"):
        assert np.array_equal(getattr(b, name), getattr(hb, name)), name
    f.close()
    recs = list(bamio.AlignmentReader(out, "rb"))
    assert len(recs) == hb.n and recs[0].qname == "r1000" and recs[5].qual is None
    assert recs[7].seq == segs[7].query_sequence and recs[7].cigar == [tuple(t) for t in segs[7].cigartuples]


def test_rank_files_join_into_one_bam_and_shares_balance(tmp_path, monkeypatch):
    """A multi-rank run writes ONE trimmed BAM like AmpliPy.py does (A:326-356, A:911): every rank re-encodes the rows of its
    share into a file of its own, rank 0 joins them block-wise (stitch_bam_parts: part 0 without its EOF block, the others
    without header blocks and EOF block).  For 2, 3 and 8 ranks the joined file inflates to the same bytes as the file one
    writer makes of all rows, ends with one EOF block and reads back record by record.  And the cut by compressed bytes
    (native_parts / ampbam_open_range) gives ranks equal work on input of the config-5 kind (mixed 75-300 bp reads, soft clips,
    indel-heavy CIGARs): bases per rank within 10 % of the mean."""
    import gzip as gz
    from amplipy_amd import amplipy
    from tools.e2e_legs import write_bam
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    hb = synth.make_config5_batch(g, amps, rep=6, pool_reads=20000)            # 120,000 reads of the config-5 mix
    seed = str(tmp_path / "seed.bam")
    write_bam(seed, synth.make_amplicon_batch(g, amps, 8, seed=1), int(g.size))
    like = bam_native.BamFile(seed)
    inp = str(tmp_path / "in.bam")
    w = bam_native.BamWriter(inp, like.header_text, like, level=6)
    w.write_batch(hb)
    w.close()
    rng = np.random.default_rng(8)

    def rows_of(piece, writer):
        b, _ = piece.decode(0, piece.n_records, copy=True)
        keep = (b.pos % 7 != 0).astype(np.uint8)                                 # some rows dropped, like the write filter does
        writer.write_rows(piece, b.src_index, keep, b.pos + 1, np.diff(b.cig_off.astype(np.int64)).astype(np.uint32), b.cig_off[:-1], b.cig)
        return int(b.lseq.sum(dtype=np.int64))

    monkeypatch.setenv("AMPLIPY_PART_BYTES", "400000")
    whole = str(tmp_path / "whole.bam")
    src = amplipy.NativeInput(inp, 0, 1)
    first = src.first_part()
    hdr_text = first.header_text
    w = bam_native.BamWriter(whole, hdr_text, first, level=1)
    for piece in src:
        rows_of(piece, w); piece.close()
    w.close()
    want = gz.decompress(open(whole, "rb").read())
    for world in (2, 3, 8):
        parts, bases = [], []
        for rank in range(world):
            src = amplipy.NativeInput(inp, rank, world)
            first = src.first_part()
            path = str(tmp_path / ("w%d.part%d.bam" % (world, rank)))
            w = bam_native.BamWriter(path, hdr_text, first, level=1)
            nb = 0
            for piece in src:
                nb += rows_of(piece, w); piece.close()
            w.close()
            parts.append((path, w.header_bytes)); bases.append(nb)
        out = str(tmp_path / ("joined%d.bam" % world))
        bam_native.stitch_bam_parts(out, parts)
        blob = open(out, "rb").read()
        assert blob.endswith(bam_native.BGZF_EOF) and blob.count(bam_native.BGZF_EOF) >= 1
        assert gz.decompress(blob) == want, world
        f = bam_native.BamFile(out)
        assert f.n_records == len(bamio_records(whole))
        f.close()
        mean = sum(bases) / world
        assert sum(bases) == int(hb.lseq.sum(dtype=np.int64))
        assert max(abs(b - mean) for b in bases) <= 0.10 * mean, (world, bases)
    like.close()


def bamio_records(path):
    return list(bamio.AlignmentReader(path, "rb"))
