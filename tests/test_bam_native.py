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
