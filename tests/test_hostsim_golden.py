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
