"""Host-side pieces of the drop-in surface that need no GPU: SAM/BAM codecs, header @PG rule,
loaders, CLI flags."""
import gzip
import os

import numpy as np
import pytest

from amplipy_amd import amplipy, bamio, synth
from amplipy_amd.batch import ReadBatch
from tests import helpers as H


def _make_sam(path, n=300):
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    segs = synth.make_mixed_segments(g, amps, n, seed=9)
    with open(path, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:SYN_REF\tLN:%d\n@PG\tID:sim\tPN:sim\tVN:1\tCL:sim x\n" % g.size)
        for i, s in enumerate(segs):
            q = "".join(chr(v + 33) for v in s.query_qualities)
            f.write("r%d\t%d\tSYN_REF\t%d\t60\t%s\t=\t%d\t%d\t%s\t%s\tNM:i:%d\tXZ:Z:abc\tXB:B:c,1,-2,3\n" % (
                i, s.flag, s.reference_start + 1, s.cigarstring, s.reference_start + 1, s.template_length,
                s.query_sequence, q, i % 7))
    return segs


def test_sam_bam_roundtrip(tmp_path):
    sam = str(tmp_path / "a.sam"); bam = str(tmp_path / "a.bam"); sam2 = str(tmp_path / "b.sam")
    segs = _make_sam(sam)
    rd = bamio.AlignmentReader(sam, "r")
    recs = list(rd)
    assert len(recs) == len(segs) and rd.header.refs == [("SYN_REF", 29903)]
    w = bamio.AlignmentWriter(bam, "wb", rd.header)
    for r in recs:
        w.write(r)
    w.close()
    rb = bamio.AlignmentReader(bam, "rb")
    recs_b = list(rb)
    assert rb.header.text == rd.header.text and rb.header.refs == rd.header.refs
    w2 = bamio.AlignmentWriter(sam2, "w", rb.header)
    for r in recs_b:
        w2.write(r)
    w2.close()
    assert open(sam).read() == open(sam2).read()
    for a, b, s in zip(recs, recs_b, segs):
        assert (a.qname, a.flag, a.pos, a.cigar, a.tlen, a.seq, a.qual) == (b.qname, b.flag, b.pos, b.cigar, b.tlen, b.seq, b.qual)
        assert a.pos == s.reference_start and a.cigar == s.cigartuples and a.seq == s.query_sequence
    # gzip-compatible container with the BGZF EOF marker
    raw = open(bam, "rb").read()
    assert raw.endswith(bamio.BGZF_EOF) and gzip.decompress(raw)[:4] == b"BAM\1"


def test_pg_header_rule():
    h = bamio.Header("@HD\tVN:1.6\n@PG\tID:bwa\tPN:bwa\n", [])
    h2 = h.with_amplipy_pg("0.0.2", "AmpliPy.py trim")
    assert h2.text.splitlines()[-1] == "@PG\tID:AmpliPy\tPN:AmpliPy\tPP:bwa\tVN:0.0.2\tCL:AmpliPy.py trim"
    h3 = h2.with_amplipy_pg("0.0.2", "x")
    assert "ID:AmpliPy.1\tPN:AmpliPy\tPP:AmpliPy" in h3.text
    with pytest.raises(KeyError):
        bamio.Header("@HD\tVN:1.6\n", []).with_amplipy_pg("0.0.2", "x")


def test_loaders_on_example_data():
    rid, seq = amplipy.load_ref_genome(H.GOLDEN + "/data/example_reference.fas")
    assert rid == "NC_045512.2" and len(seq) == 29903
    pr = amplipy.load_primers(H.GOLDEN + "/data/example_primers.bed")
    assert len(pr) == 690 and pr == sorted(pr) and max(e - s for s, e in pr) == 30


def test_cli_flags_and_defaults():
    a = amplipy.parse_args(["aio", "-p", "p.bed", "-r", "r.fas", "-ot", "t.bam", "-ov", "v.vcf", "-oc", "c.fas"])
    assert (a.input, a.primer_pos_offset, a.min_length, a.min_quality, a.sliding_window_width) == ("stdin", 0, 30, 20, 4)
    assert (a.min_freq_consensus, a.min_freq_variants, a.min_depth_consensus, a.min_depth_variants) == (0, 0.03, 10, 1)
    assert a.unknown_symbol == "N" and a.include_no_primer is False
    t = amplipy.parse_args(["trim", "-p", "p", "-r", "r", "-x", "5", "-ml", "40", "-mq", "25", "-s", "7", "-e"])
    assert (t.output, t.primer_pos_offset, t.min_length, t.min_quality, t.sliding_window_width, t.include_no_primer) == \
        ("stdout", 5, 40, 25, 7, True)
    v = amplipy.parse_args(["variants", "-r", "r"]); c = amplipy.parse_args(["consensus", "-r", "r"])
    assert (v.min_freq, v.min_depth, c.min_freq, c.min_depth, c.unknown_symbol) == (0.03, 1, 0, 10, "N")


def test_argument_validation_exits_like_the_reference(capsys):
    with pytest.raises(SystemExit) as e:
        amplipy.run_amplipy(min_quality=-1, run_trim=True)
    assert e.value.code == 1
    assert "ERROR: Minimum quality must be non-negative: -1" in capsys.readouterr().err
    with pytest.raises(SystemExit):
        amplipy.run_amplipy()
    assert "Not running any of the AmpliPy operations" in capsys.readouterr().err


def test_event_store_matches_per_event_strings():
    """EventStore (vectorised, what run_amplipy keeps) against insertions.event_strings (per event)."""
    from amplipy_amd import insertions
    from oracle import oracle
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    segs = synth.make_mixed_segments(g, amps, 3000, seed=8)
    b = ReadBatch.from_segments(segs)
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, [(s, e) for s, e, _ in primers], 0)
    r = oracle.process(b, g.size, mn, mx, mpl, 20, 4, read_base=1000)
    assert r.events.size > 500
    st = insertions.EventStore()
    half = r.events.size // 2
    st.add(b, r.events[:half], 1000); st.add(b, r.events[half:], 1000); st.add(b, r.events[:0], 0)
    want = insertions.event_strings(b, r.events, 1000)
    assert st.pairs() == want and len(st) == len(want)
    some = set(p for p, _ in want[::7])
    assert st.pairs(some) == [x for x in want if x[0] in some]
    assert st.pairs(set()) == []


def test_event_store_with_run_counts_and_tallies():
    """Rows that stand for several events (the (position, allele) runs of amp_aggregate_ins_events): the store
    keeps their counts, and calling.tallies_from_runs sums rows of one allele (a run may be split, never mixed)."""
    from collections import Counter
    from amplipy_amd import calling, insertions
    st = insertions.EventStore()
    st.add_text([5, 5, 9], [2, 3, 2], np.frombuffer(b"ATATTAC", np.uint8), [4, 1, 2])
    st.add_text([5], [2], np.frombuffer(b"AT", np.uint8), [3])                 # the same allele again (another batch, or a split run)
    assert len(st) == 10
    assert st.counted_pairs() == [(5, "AT", 4), (5, "ATT", 1), (9, "AC", 2), (5, "AT", 3)]
    assert Counter(st.pairs()) == Counter({(5, "AT"): 7, (5, "ATT"): 1, (9, "AC"): 2})
    t = calling.tallies_from_runs(st.counted_pairs({5}), {5})
    assert dict(t[5]) == {"AT": 7, "ATT": 1} and 9 not in t
    assert calling.tallies_from_runs(st.counted_pairs(), {9})[9] == {"AC": 2}


def test_vcf_text_equals_the_record_path():
    """CallResult.vcf_text (the VCF body straight from the columns, what run_amplipy writes) is the text VcfWriter.line gives
    for every VariantRecord of ``records`` -- positions finished from insertion alleles (``extra``: a replaced record, a
    dropped one) and reference counts of zero included."""
    import numpy as np
    from amplipy_amd import calling
    from amplipy_amd.amplipy import VcfWriter
    V, G = 3000, 29903
    rng = np.random.default_rng(1)
    pos = np.sort(rng.choice(G, V, replace=False)).astype(np.int32)
    extra = {int(pos[5]): None,
             int(pos[9]): calling.VariantRecord(int(pos[9]), "A", ["ACG", "T"], 100, 3, [50, 47], 0.03, [0.5, 0.47], (0, 1, 2)),
             17: calling.VariantRecord(17, "C", ["CTT"], 40, 0, [40], 0, [1.0], (1,))}
    res = calling.CallResult("ACGT" * 8000, np.zeros(G, np.int8), {}, pos, rng.integers(100, 10000, V).astype(np.uint32),
                             rng.integers(0, 100, V).astype(np.uint32), rng.integers(0, 2, V).astype(bool),
                             rng.integers(1, 4, V).astype(np.int8), rng.integers(0, 6, (V, 6)).astype(np.int8),
                             rng.integers(1, 50, (V, 6)).astype(np.uint32), extra, None, 0)
    res.var_ref_count[:50] = 0

    class W(VcfWriter):
        def __init__(self):
            self.ref_id = "REF"
    w = W()
    assert res.vcf_text("REF") == "".join(w.line(r) for r in res.records)
