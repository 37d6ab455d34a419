"""Pins the CPU restatement (oracle/) to the reference-derived golden vectors (tests/golden)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


def _oracle_process(b, G, mn, mx, mpl, mq, w, do_trim):
    return oracle.process(b, G, mn, mx, mpl, mq, w, do_trim=do_trim, do_count=True)


def test_primer_tables_example_and_random():
    meta = H.load_json("primer_tables.json")
    tabs = np.load(H.GOLDEN + "/primer_tables.npz")
    bed = [l.rstrip("\r\n").split("\t") for l in open(H.GOLDEN + "/data/example_primers.bed") if l.strip()]
    primers = [(int(f[1]), int(f[2])) for f in bed]
    assert len(primers) == meta["example"]["n_primers"] == 690
    for off, covered in ((0, 15026), (5, 20752)):
        mn, mx, mpl = oracle.find_overlapping_primers(meta["example"]["ref_len"], primers, off)
        assert mpl == 30
        assert np.array_equal(mn, tabs["example_off%d_min_start" % off])
        assert np.array_equal(mx, tabs["example_off%d_max_end" % off])
        assert int((mn >= 0).sum()) == covered
    for k, s in enumerate(meta["random_sets"]):
        mn, mx, mpl = oracle.find_overlapping_primers(s["ref_len"], s["primers"], s["offset"])
        assert mpl == s["max_primer_len"]
        assert np.array_equal(mn, tabs["rand%d_min_start" % k])
        assert np.array_equal(mx, tabs["rand%d_max_end" % k])


def test_coordinate_helpers():
    for c in H.load_json("helpers.json")["cases"]:
        cig = [tuple(x) for x in c["cigar"]]
        assert oracle.pos_on_query(cig, c["start"] + c["x"], c["start"]) == (c["pos_on_query"], 0)
        assert oracle.pos_on_ref(cig, c["x"], c["start"]) == (c["pos_on_ref"], 0)
        assert oracle.fix_cigar(cig) == [tuple(x) for x in c["fix_cigar"]]


@pytest.mark.parametrize("fixture", ["named_cases.json", "random_reads.json.gz"])
def test_trim_and_count_per_read(fixture):
    fails = []
    for case in H.load_json(fixture)["cases"]:
        fails += H.check_case_with(_oracle_process, case, oracle.find_overlapping_primers)
    assert not fails, "\n".join(fails[:40])


def test_known_answers_example_reads():
    """the two example SAM reads."""
    case = H.load_json("named_cases.json")["cases"][0]
    assert case["name"] == "example_reads"
    e0, e1 = case["expected"]
    assert e0["trim"] == {"pos": 26, "cigar": "24S51M76H", "flags": [True, False, False], "reflen": 51}
    assert e1["trim"] == {"pos": 28254, "cigar": "31S105M15S", "flags": [False, True, False], "reflen": 105}
    assert len(e0["count_trimmed"]["counts"]) == 51 and len(e1["count_trimmed"]["counts"]) == 103


def test_pileup_batch_counts():
    g = H.load_json("pileup_5000.json.gz")
    from amplipy_amd.batch import ReadBatch
    b = ReadBatch.from_segments([H.seg_from_dict(d) for d in g["reads"]])
    mn, mx, mpl = oracle.find_overlapping_primers(g["ref_len"], g["primers"], g["offset"])
    assert mpl == g["max_primer_len"]
    r = oracle.process(b, g["ref_len"], mn, mx, mpl, g["params"]["min_quality"], g["params"]["window"])
    assert not r.trim.status.any()
    for i, (pos, cig, flags, reflen) in enumerate(g["trim"]):
        assert (int(r.trim.new_pos[i]), r.trim.cigar_string(i), int(r.trim.ref_len[i])) == (pos, cig, reflen), i
        assert [bool(r.trim.trim_flags[i] & m) for m in (1, 2, 4)] == flags
    assert H.sparse_from_engine(r.counts, b, r.events) == H.sparse_from_golden(g["counts"])


def test_allele_ranking():
    g = H.load_json("pileup_5000.json.gz")
    from collections import defaultdict
    per_pos = defaultdict(dict)
    for p, k, n in g["counts"]:
        per_pos[p][k] = n
    for rec in g["calls"]:
        d = per_pos[rec["pos"]]
        syms = list("ACGTN-") + sorted(k for k in d if k not in tuple("ACGTN-"))
        cnt = [d.get(s, 0) for s in syms]
        total, order = oracle.rank_alleles(syms, cnt)
        assert total == rec["total"]
        assert [[cnt[i], syms[i]] for i in order] == [[a[0], a[2]] for a in rec["alleles"]]


def test_python_restatement_equals_c_oracle_and_golden():
    """oracle/py_restatement.py (the CPU-baseline leg with AmpliPy's own algorithmic shape) against the C oracle
    and the reference-derived pileup fixture."""
    from amplipy_amd import synth
    from amplipy_amd.batch import ReadBatch
    from oracle import py_restatement
    g = H.load_json("pileup_5000.json.gz")
    reads = [H.seg_from_dict(d) for d in g["reads"]]
    b = ReadBatch.from_segments(reads)
    mn, mx, mpl = oracle.find_overlapping_primers(g["ref_len"], g["primers"], g["offset"])
    counts, events, trims = py_restatement.process_full(b, g["ref_len"], mn, mx, mpl, g["params"]["min_quality"], g["params"]["window"])
    ref = oracle.process(b, g["ref_len"], mn, mx, mpl, g["params"]["min_quality"], g["params"]["window"])
    assert np.array_equal(counts, ref.counts)
    assert sorted(events) == sorted((int(e["ref_pos"]), int(e["read"]), int(e["q_from"]), int(e["q_to"])) for e in ref.events)
    from amplipy_amd.segment import format_cigar
    for i, (pos, cig, flags, reflen) in enumerate(g["trim"]):
        assert (trims[i][0], format_cigar(trims[i][1]), list(trims[i][2])) == (pos, cig, flags), i
    # mixed shapes (soft clips, indel-heavy CIGARs), no trimming
    genome = synth.make_genome(); _, amps = synth.make_artic_scheme()
    b2 = ReadBatch.from_segments(synth.make_mixed_segments(genome, amps, 600, seed=12))
    none = np.full(genome.size, -1, np.int32)
    c2 = py_restatement.process_full(b2, genome.size, none, none, 0, 20, 4, do_trim=False)[0]
    assert np.array_equal(c2, oracle.process(b2, genome.size, do_trim=False).counts)
