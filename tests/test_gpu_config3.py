"""BASELINE config 3 at its real size, as ONE launch: 100k x depth = 19,935,333 reads of 150 bp (3.03 GB of
qualities: byte offsets beyond 2^31) through amp_process_batch_device, every per-read output, the count table
and the insertion events compared with the CPU oracle (sharded over threads; A:896-915 is per read, the table is
an order-free integer sum)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from amplipy_amd import abi, lib, synth
from oracle import oracle

pytestmark = pytest.mark.gpu

EV_ORDER = ["ref_pos", "read", "q_from", "q_to"]


def test_config3_single_launch_equals_oracle():
    import torch
    from amplipy_amd import synth_torch
    g = synth.make_genome()
    primers, amps = synth.make_artic_scheme()
    G = int(g.size)
    mn, mx, mpl = oracle.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
    n = synth.reads_for_depth(100000)
    assert n == 19935333
    b = synth_torch.make_amplicon_batch_device(g, amps, n, seed=2, device="cuda:0")
    assert b.n_bases_padded > 2 ** 31          # the point of the test: 64-bit byte offsets in one launch
    e = lib.Engine(G)
    e.set_primers(mn, mx, mpl)
    e.set_params(20, 4, True, True)
    e.reserve_events(n // 4)
    out = {k: torch.zeros(sz, dtype=dt, device="cuda:0") for k, sz, dt in
           (("new_pos", n, torch.int32), ("new_ncig", n, torch.int32), ("new_cig", b.n_cig + 3 * n, torch.int32),
            ("ref_len", n, torch.int32), ("trim_flags", n, torch.uint8), ("status", n, torch.uint8))}
    dev_out = abi.AmpTrimOut(*[out[k].data_ptr() for k in ("new_pos", "new_ncig", "new_cig", "ref_len", "trim_flags", "status")])
    e.process_device(b.struct(), 0, dev_out)
    e.sync()
    counts = e.counts()
    events = e.events()
    h = {k: v.cpu().numpy() for k, v in out.items()}
    new_cig = h["new_cig"].view(np.uint32)
    hb = b.to_host()
    del b, out
    torch.cuda.empty_cache()

    shards = 16
    cuts = [n * k // shards for k in range(shards + 1)]

    def shard(k):
        lo, hi = cuts[k], cuts[k + 1]
        sb = hb.slice(lo, hi)
        r = oracle.process(sb, G, mn, mx, mpl, 20, 4, read_base=lo)
        t = r.trim
        assert not t.status.any() and not h["status"][lo:hi].any(), "shard %d: a read reports an error" % k
        assert np.array_equal(t.new_pos, h["new_pos"][lo:hi]), "shard %d: new_pos" % k
        assert np.array_equal(t.new_ncig, h["new_ncig"][lo:hi].view(np.uint32)), "shard %d: new_ncig" % k
        assert np.array_equal(t.ref_len, h["ref_len"][lo:hi]), "shard %d: ref_len" % k
        assert np.array_equal(t.trim_flags, h["trim_flags"][lo:hi]), "shard %d: trim_flags" % k
        # the new CIGAR of read i sits at cig_off[i] + 3 i of the launch's slot array
        nc = t.new_ncig.astype(np.int64)
        start = hb.cig_off[lo:hi].astype(np.int64) + 3 * np.arange(lo, hi, dtype=np.int64)
        first = np.repeat(start - (np.cumsum(nc) - nc), nc)
        got = new_cig[first + np.arange(int(nc.sum()), dtype=np.int64)]
        assert np.array_equal(got, t.compact_cigars()), "shard %d: CIGAR words" % k
        return r.counts, r.events

    with ThreadPoolExecutor(shards) as pool:
        parts = list(pool.map(shard, range(shards)))
    total = np.sum([p[0] for p in parts], axis=0, dtype=np.uint32)
    assert np.array_equal(total, counts), "count table differs from the oracle"
    ref_events = np.concatenate([p[1] for p in parts])
    assert ref_events.size == events.size
    assert np.array_equal(np.sort(ref_events, order=EV_ORDER), np.sort(events, order=EV_ORDER)), "insertion events differ"
    e.close()
