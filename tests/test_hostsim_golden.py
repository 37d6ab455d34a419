"""The per-read device functions (amp_read.hpp), executed on the CPU by tests/hostsim, against
the reference-derived golden vectors and against the oracle on random reads."""
import shutil

import numpy as np
import pytest

from amplipy_amd import synth
from amplipy_amd.batch import ReadBatch
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc needed to build hostsim")


def _sim(b, G, mn, mx, mpl, mq, w, do_trim):
    from tests import hostsim
    return hostsim.process(b, G, mn, mx, mpl, mq, w, do_trim=do_trim)


@pytest.mark.parametrize("fixture", ["named_cases.json", "random_reads.json.gz"])
def test_device_read_logic_matches_reference(fixture):
    fails = []
    for case in H.load_json(fixture)["cases"]:
        fails += H.check_case_with(_sim, case, oracle.find_overlapping_primers)
    assert not fails, "\n".join(fails[:40])


@pytest.mark.parametrize("seed,mq,w", [(1, 20, 4), (2, 0, 1), (3, 30, 9), (4, 25, 200)])
def test_device_read_logic_matches_oracle_random(seed, mq, w):
    from tests import hostsim
    g = synth.make_genome()
    primers, _ = synth.make_artic_scheme()
    pr = [(s, e) for s, e, _ in primers]
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, seed % 3)
    rng = np.random.default_rng(seed)
    segs = synth.random_segments(rng, 3000, g.size, pr)
    for s in segs:  # one read per batch so an error read cannot disturb its neighbours
        pass
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, g.size, mn, mx, mpl, mq, w)
    d = hostsim.process(b, g.size, mn, mx, mpl, mq, w)
    assert np.array_equal(a.trim.status, d.trim.status)
    ok = a.trim.status == 0
    assert np.array_equal(a.trim.new_pos[ok], d.trim.new_pos[ok])
    assert np.array_equal(a.trim.ref_len, d.trim.ref_len)
    assert np.array_equal(a.trim.trim_flags, d.trim.trim_flags)
    assert np.array_equal(a.trim.new_ncig, d.trim.new_ncig)
    for i in np.nonzero(ok)[0]:
        assert a.trim.cigar_ops(i) == d.trim.cigar_ops(i)
    # counts: compare on the error-free subset only
    good = ReadBatch.from_segments([segs[i] for i in np.nonzero(ok)[0]])
    a = oracle.process(good, g.size, mn, mx, mpl, mq, w)
    d = hostsim.process(good, g.size, mn, mx, mpl, mq, w)
    assert not a.trim.status.any() and not d.trim.status.any()
    assert np.array_equal(a.counts, d.counts)
    assert np.array_equal(np.sort(a.events, order=["ref_pos", "read", "q_from", "q_to"]),
                          np.sort(d.events, order=["ref_pos", "read", "q_from", "q_to"]))


@pytest.mark.parametrize("seed,mq,w,off", [(11, 20, 4, 0), (12, 20, 4, 7), (13, 35, 3, 3), (14, 2, 8, 1), (15, 20, 40, 0)])
def test_closed_form_trims_of_single_op_reads(seed, mq, w, off):
    """Reads whose CIGAR is one match op take closed-form trims on the device (amp_read.hpp SimpleCig);
    the oracle walks the general loops.  Dense random coverage of their cases: reads starting / ending inside,
    before and after primers (offsets make `del` negative), whole-read clips, every strand / pairing flag,
    insert sizes on both sides of the A:452 rule, low-quality runs at both ends, missing qualities."""
    from amplipy_amd.segment import Segment
    from tests import hostsim
    G = 4000
    rng = np.random.default_rng(seed)
    starts = np.sort(rng.choice(np.arange(10, G - 60), 60, replace=False))
    pr = [(int(s), int(s) + int(rng.integers(15, 31))) for s in starts]
    mn, mx, mpl = oracle.find_overlapping_primers(G, pr, off)
    segs = []
    for _ in range(12000):
        L = int(rng.integers(1, 90))
        ps, pe = pr[int(rng.integers(0, len(pr)))]
        mode = rng.random()
        if mode < 0.45:
            pos = ps + int(rng.integers(-12 - off, pe - ps + 12 + off))
        elif mode < 0.9:
            pos = pe + int(rng.integers(-12 - off, 12 + off)) - L
        else:
            pos = int(rng.integers(0, G))
        pos = min(max(pos, 0), G - L)
        q = rng.choice([37, 25, 11, 2], L, p=[0.7, 0.15, 0.1, 0.05])
        if rng.random() < 0.3:
            k = int(rng.integers(0, L + 1)); q[:k] = 2
        if rng.random() < 0.3:
            k = int(rng.integers(0, L + 1)); q[L - k:] = 2
        quals = None if rng.random() < 0.02 else q.tolist()
        tl = int(rng.choice([0, L, L + mpl, L + mpl + 1, 400, -400, -(L + mpl + 1), -L]))
        segs.append(Segment(flag=int(rng.choice([0, 16, 1, 17, 99, 147, 83, 163])), reference_start=pos,
                            cigar=[(int(rng.choice([0, 7, 8])), L)], template_length=tl,
                            query_sequence="".join(rng.choice(list("ACGT"), L)), query_qualities=quals))
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, mpl, mq, w, do_count=False)
    d = hostsim.process(b, G, mn, mx, mpl, mq, w, do_count=False)
    assert np.array_equal(a.trim.status, d.trim.status)
    ok = a.trim.status == 0
    assert ok.mean() > 0.9 and (~ok).any()
    assert np.array_equal(a.trim.new_pos[ok], d.trim.new_pos[ok])
    assert np.array_equal(a.trim.ref_len, d.trim.ref_len)
    assert np.array_equal(a.trim.trim_flags, d.trim.trim_flags)
    assert np.array_equal(a.trim.new_ncig, d.trim.new_ncig)
    assert np.array_equal(a.trim.compact_cigars(), d.trim.compact_cigars())
    assert len(set(a.trim.trim_flags.tolist())) >= 4          # the cases are really visited


def test_two_segment_closed_forms():
    """The closed forms for reads with at most one insertion / deletion (Cig2 in amp_read.hpp: primer clips, quality
    clip, counted segments, deletion counts, insertion events) against the generic exact code on random reads,
    primer tables, windows and qualities.  Cases the closed forms do not cover must be flagged (punt), never wrong."""
    from tests import hostsim
    for seed in (11, 12, 13):
        bad, punted, compared = hostsim.cig2_fuzz(seed, 200000)
        assert bad == 0, (seed, bad)
        assert compared > 190000 and punted < 5000


def test_branch_free_closed_forms():
    """amp_bf.hpp (the closed forms of Cig2 written as arithmetic and selects: what the fast kernel, variant 5, runs) against
    the branchy forms of the test above on random shapes, table entries aimed at and around the indel, both strands and
    EVERY outcome of the window scan: parse result, trimmed shape, position and flags must be identical."""
    from tests import hostsim
    for seed in (21, 22, 23):
        bad, punted, compared = hostsim.bf_fuzz(seed, 150000)
        assert bad == 0, (seed, bad)
        assert compared > 1000000


def test_op_by_op_indel_walk():
    """count_regular_ops (amp_read.hpp: deletions and insertion events of a regular CIGAR op by op, what the tile kernel's
    lanes run) against the exact pair walk count_read_walk on random regular CIGARs -- zero-length ops, insertions next to
    each other, next to deletions and clips, at reference position 0, over the reference's end, low bases inside the
    insertions: the same '-' counts, the same events in the same order, the same error / no error outcome."""
    from tests import hostsim
    for seed in (31, 32):
        bad, effects, errors = hostsim.regops_fuzz(seed, 200000)
        assert bad == 0, (seed, bad)
        assert effects > 1000000 and errors > 10000
