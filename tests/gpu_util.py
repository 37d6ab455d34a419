"""GPU-side test plumbing: an Engine wrapper with the same call shape as oracle.process."""
import numpy as np

from amplipy_amd import lib

EV_ORDER = ["ref_pos", "read", "q_from", "q_to"]


class Result:
    def __init__(self, trim, counts, events):
        self.trim = trim; self.counts = counts; self.events = events


class GpuRunner:
    """Keeps one Engine per reference length; process() mirrors oracle.process()."""

    def __init__(self, variant=None):
        self.engines = {}
        self.variant = variant

    def engine(self, ref_len):
        if ref_len not in self.engines:
            e = lib.Engine(ref_len)
            if self.variant is not None:
                e.set_kernel_variant(self.variant)
            self.engines[ref_len] = e
        return self.engines[ref_len]

    def process(self, batch, ref_len, mn=None, mx=None, mpl=0, mq=20, w=4, do_trim=True, do_count=True,
                read_base=0, reset=True):
        e = self.engine(ref_len)
        if reset:
            e.reset()
        if mn is None:
            mn = np.full(ref_len, -1, np.int32); mx = mn
        e.set_primers(mn, mx, mpl)
        e.set_params(mq, w, do_trim, do_count)
        trim = e.process(batch, read_base=read_base)
        return Result(trim, e.counts(), e.events())

    def close(self):
        for e in self.engines.values():
            e.close()
        self.engines = {}


def assert_same(a, d, batch, check_counts=True):
    """a = oracle result, d = device result (both over the same batch)."""
    assert np.array_equal(a.trim.status, d.trim.status), "status differs"
    ok = a.trim.status == 0
    assert np.array_equal(a.trim.new_pos[ok], d.trim.new_pos[ok]), "new_pos differs"
    assert np.array_equal(a.trim.ref_len, d.trim.ref_len), "ref_len differs"
    assert np.array_equal(a.trim.trim_flags, d.trim.trim_flags), "trim_flags differ"
    assert np.array_equal(a.trim.new_ncig, d.trim.new_ncig), "new_ncig differs"
    assert np.array_equal(a.trim.compact_cigars(), d.trim.compact_cigars()), "new CIGARs differ"
    if check_counts:
        assert not a.trim.status.any()
        assert np.array_equal(a.counts, d.counts), "count table differs"
        assert np.array_equal(np.sort(a.events, order=EV_ORDER), np.sort(d.events, order=EV_ORDER)), "events differ"
