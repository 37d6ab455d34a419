"""BASELINE config 5 at its real size, as ONE launch: 50k x depth = 8.0 M mixed 75-300 bp reads with long soft clips and
indel-heavy CIGARs (the divergent-wavefront stress; per-read driver A:896-915) through amp_process_batch, every per-read
output, the count table and the insertion events compared with the CPU oracle (sharded over threads: A:896-915 is per read,
the table is an order-free integer sum)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from amplipy_amd import lib, synth
from oracle import oracle

pytestmark = pytest.mark.gpu

EV_ORDER = ["ref_pos", "read", "q_from", "q_to"]


def test_config5_single_launch_equals_oracle():
    g = synth.make_genome()
    primers, amps = synth.make_artic_scheme()
    G = int(g.size)
    mn, mx, mpl = oracle.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
    b = synth.make_config5_batch(g, amps, 200)
    n = b.n
    assert n == 8_000_000 and 75 <= int(b.lseq.min()) and int(b.lseq.max()) <= 300
    nops = np.diff(b.cig_off.astype(np.int64))
    assert (nops > 1).mean() > 0.5                      # indel-heavy, soft-clipped
    e = lib.Engine(G)
    e.set_primers(mn, mx, mpl)
    e.set_params(20, 4, True, True)
    e.reserve_events(n // 2)
    res = e.process(b)
    counts = e.counts()
    events = e.events()
    got_cig = res.compact_cigars()
    goff = np.concatenate([[0], np.cumsum(res.new_ncig.astype(np.int64))])

    shards = 16
    cuts = [n * k // shards for k in range(shards + 1)]

    def shard(k):
        lo, hi = cuts[k], cuts[k + 1]
        r = oracle.process(b.slice(lo, hi), G, mn, mx, mpl, 20, 4, read_base=lo)
        t = r.trim
        assert np.array_equal(t.status, res.status[lo:hi]), "shard %d: status" % k
        okr = t.status == 0
        assert np.array_equal(t.new_pos[okr], res.new_pos[lo:hi][okr]), "shard %d: new_pos" % k
        assert np.array_equal(t.new_ncig, res.new_ncig[lo:hi]), "shard %d: new_ncig" % k
        assert np.array_equal(t.ref_len, res.ref_len[lo:hi]), "shard %d: ref_len" % k
        assert np.array_equal(t.trim_flags, res.trim_flags[lo:hi]), "shard %d: trim_flags" % k
        assert np.array_equal(t.compact_cigars(), got_cig[goff[lo]:goff[hi]]), "shard %d: CIGAR words" % k
        return r.counts, r.events

    with ThreadPoolExecutor(shards) as pool:
        parts = list(pool.map(shard, range(shards)))
    total = np.sum([p[0] for p in parts], axis=0, dtype=np.uint32)
    assert np.array_equal(total, counts), "count table differs from the oracle"
    ref_events = np.concatenate([p[1] for p in parts])
    assert ref_events.size == events.size and events.size > 100000
    assert np.array_equal(np.sort(ref_events, order=EV_ORDER), np.sort(events, order=EV_ORDER)), "insertion events differ"
    e.close()
