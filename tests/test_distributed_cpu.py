"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo).  The GPU kernels cannot run here,
so each rank computes its shard with the CPU oracle; what is under test is the host logic that
the bench and run_amplipy use across ranks: the coordinate-range partition, the single reduce
of the 7 x ref_len table, and the agreement / gather protocol for insertion-relevant positions."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from amplipy_amd import parallel, synth
from amplipy_amd.insertions import event_strings
from oracle import oracle


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_reads, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
        pr = [(s, e) for s, e, _ in primers]
        mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, 0)
        batch = synth.make_amplicon_batch(g, amps, n_reads, seed=77)     # same batch on every rank
        cuts = parallel.shard_bounds(batch, world)
        lo, hi = cuts[rank], cuts[rank + 1]
        mine = batch.slice(lo, hi)
        r = oracle.process(mine, g.size, mn, mx, mpl, 20, 4, read_base=lo)
        table = np.zeros(g.size * 7, np.int64)
        table[:g.size * 6] = r.counts.reshape(-1)
        np.add.at(table[g.size * 6:], r.events["ref_pos"], 1)
        t = torch.from_numpy(table.astype(np.int32))
        parallel.reduce_table(dist, t, dst=0)
        # pretend every 97th position with events is insertion-relevant
        rel = None
        if rank == 0:
            tally = t.numpy()[g.size * 6:]
            rel = [int(p) for p in np.nonzero(tally)[0][::97]]
        rel = parallel.agree_on_positions(dist, rank, rel)
        ev = r.events[np.isin(r.events["ref_pos"], rel)]
        ev_local = ev.copy(); ev_local["read"] -= lo
        pairs = event_strings(mine, ev_local)
        allpairs = parallel.gather_relevant_events(dist, rank, world, pairs)
        if rank == 0:
            np.savez(out_path, table=t.numpy(), rel=np.array(rel), pairs=np.array(sorted(allpairs), dtype=object),
                     cuts=np.array(cuts))
    finally:
        dist.destroy_process_group()


def _worker_allreduce(rank, world, port, n_reads, out_dir):
    """The bench's protocol: ONE all-reduce, then every rank decides the relevant positions from the
    same table and the insertion strings are exchanged symmetrically."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
        pr = [(s, e) for s, e, _ in primers]
        mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, 0)
        batch = synth.make_amplicon_batch(g, amps, n_reads, seed=78)
        cuts = parallel.shard_bounds(batch, world)
        lo, hi = cuts[rank], cuts[rank + 1]
        mine = batch.slice(lo, hi)
        r = oracle.process(mine, g.size, mn, mx, mpl, 20, 4, read_base=lo)
        table = np.zeros(g.size * 7, np.int64)
        table[:g.size * 6] = r.counts.reshape(-1)
        np.add.at(table[g.size * 6:], r.events["ref_pos"], 1)
        t = torch.from_numpy(table.astype(np.int32))
        parallel.allreduce_table(dist, t)
        rel = [int(p) for p in np.nonzero(t.numpy()[g.size * 6:])[0][::89]]     # same on every rank by construction
        ev = r.events[np.isin(r.events["ref_pos"], rel)]
        ev_local = ev.copy(); ev_local["read"] -= lo
        allpairs = parallel.allgather_relevant_events(dist, world, event_strings(mine, ev_local))
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), table=t.numpy(), rel=np.array(rel),
                 pairs=np.array(sorted(allpairs), dtype=object))
    finally:
        dist.destroy_process_group()


def test_allreduce_protocol_gives_every_rank_the_whole_result(tmp_path):
    world, n_reads = 2, 5000
    mp.spawn(_worker_allreduce, args=(world, _free_port(), n_reads, str(tmp_path)), nprocs=world, join=True)
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    pr = [(s, e) for s, e, _ in primers]
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, 0)
    batch = synth.make_amplicon_batch(g, amps, n_reads, seed=78)
    ref = oracle.process(batch, g.size, mn, mx, mpl, 20, 4)
    tally = np.zeros(g.size, np.int64); np.add.at(tally, ref.events["ref_pos"], 1)
    for rank in range(world):
        got = np.load(str(tmp_path / ("r%d.npz" % rank)), allow_pickle=True)
        table = got["table"].view(np.uint32)
        assert np.array_equal(table[:g.size * 6].reshape(g.size, 6), ref.counts)
        assert np.array_equal(table[g.size * 6:], tally.astype(np.uint32))
        rel = set(int(p) for p in got["rel"])
        assert rel, "the sample should have insertion events"
        want = sorted(p for p in event_strings(batch, ref.events) if p[0] in rel)
        assert [tuple(x) for x in got["pairs"]] == want


@pytest.mark.parametrize("world", [2])
def test_sharded_reduce_matches_single_process(tmp_path, world):
    n_reads = 6000
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_reads, out), nprocs=world, join=True)
    got = np.load(out, allow_pickle=True)
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    pr = [(s, e) for s, e, _ in primers]
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, 0)
    batch = synth.make_amplicon_batch(g, amps, n_reads, seed=77)
    ref = oracle.process(batch, g.size, mn, mx, mpl, 20, 4)
    table = got["table"].view(np.uint32)
    assert np.array_equal(table[:g.size * 6].reshape(g.size, 6), ref.counts)
    tally = np.zeros(g.size, np.int64); np.add.at(tally, ref.events["ref_pos"], 1)
    assert np.array_equal(table[g.size * 6:], tally.astype(np.uint32))
    rel = set(int(p) for p in got["rel"])
    want = sorted(p for p in event_strings(batch, ref.events) if p[0] in rel)
    assert [tuple(x) for x in got["pairs"]] == want
    cuts = got["cuts"]
    assert cuts[0] == 0 and cuts[-1] == batch.n and all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))


def test_shard_bounds_balance_bases():
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    from amplipy_amd.batch import ReadBatch
    segs = synth.make_mixed_segments(g, amps, 3000, seed=3)
    b = ReadBatch.from_segments(segs)
    for world in (2, 4, 8):
        cuts = parallel.shard_bounds(b, world)
        bases = [int(b.lseq[a:c].sum()) for a, c in zip(cuts[:-1], cuts[1:])]
        assert sum(bases) == b.total_bases()
        assert max(bases) - min(bases) <= 2 * 300      # within a couple of reads of each other
    assert parallel.amplicon_range(98, 0, 8) == (0, 12) and parallel.amplicon_range(98, 7, 8)[1] == 98


def _worker_file_partition(rank, world, port, bam_path, out_dir, fail_rank):
    """What run_amplipy does with a BAM file under torchrun, with the CPU oracle standing in for the GPU engine only: the
    rank's share comes from amplipy.NativeInput (pieces of ampbam_open_range, cut by compressed bytes at BGZF block starts,
    the next piece inflated ahead), seams and errors go through parallel.exchange_notes, ONE all-reduce stitches the tables,
    insertion alleles travel as (position, text, count) runs through parallel.allgather_relevant_events and
    calling.tallies_from_runs."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["AMPLIPY_PART_BYTES"] = str(96 << 10)          # several pieces per rank on a small file
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from collections import Counter
        from amplipy_amd import amplipy, calling
        g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
        mn, mx, mpl = oracle.find_overlapping_primers(g.size, [(s, e) for s, e, _ in primers], 0)
        from amplipy_amd import bam_native
        src = amplipy.NativeInput(bam_path, rank, world)
        first = src.first_part()
        wr = bam_native.BamWriter(os.path.join(out_dir, "trim.part%d.bam" % rank), first.header_text, first, level=1)
        table = np.zeros(g.size * 7, np.int64)
        runs = Counter()
        n_rows = n_pieces = n_bases = 0
        err = None
        try:
            for piece in src:
                n_pieces += 1
                if piece.n_records:
                    b, _ = piece.decode(0, piece.n_records, copy=True)
                    r = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
                    # the rank's trimmed reads, with the write filter of AmpliPy.py:910, into the rank's own file
                    keep = (r.trim.ref_len >= 30) & ((r.trim.trim_flags & 3) != 0)
                    slot = b.cig_off[:-1] + np.uint64(3) * np.arange(b.n, dtype=np.uint64)
                    wr.write_rows(piece, b.src_index, keep, r.trim.new_pos, r.trim.new_ncig, slot, r.trim.new_cig)
                    n_bases += int(b.lseq.sum(dtype=np.int64))
                    table[:g.size * 6] += r.counts.reshape(-1)
                    np.add.at(table[g.size * 6:], r.events["ref_pos"], 1)
                    runs.update(event_strings(b, r.events))
                    n_rows += b.n
                piece.close()
            if rank == fail_rank:
                raise ValueError("made-up failure of one rank")
        except Exception as e:
            err = e
        wr.close()
        trouble = parallel.exchange_notes(dist, world, getattr(src, "seam", [None, None]), err)
        if trouble:
            with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
                f.write(trouble)
            return
        # ONE trimmed BAM, like the single-process run writes: rank 0 joins the ranks' files (run_amplipy does exactly this)
        parts = parallel.gather_objects(dist, rank, world, (wr.path, wr.header_bytes))
        shares = parallel.gather_objects(dist, rank, world, (n_rows, n_bases))
        if rank == 0:
            bam_native.stitch_bam_parts(os.path.join(out_dir, "trim.bam"), parts)
            with open(os.path.join(out_dir, "shares.txt"), "w") as f:
                f.write(repr(shares))
        t = torch.from_numpy(table.astype(np.int32))
        parallel.allreduce_table(dist, t)
        rel = set(int(p) for p in np.nonzero(t.numpy()[g.size * 6:])[0][::53])        # same on every rank by construction
        triples = [(p, s_, c) for (p, s_), c in runs.items() if p in rel]
        tallies = calling.tallies_from_runs(parallel.allgather_relevant_events(dist, world, triples), rel)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), table=t.numpy(), n_rows=n_rows, n_pieces=n_pieces,
                 tallies=np.array(sorted((p, s_, c) for p, d in tallies.items() for s_, c in d.items()), dtype=object))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_run_amplipys_file_partition_and_gather(tmp_path, world):
    from collections import Counter
    from amplipy_amd import bam_native
    from tools.e2e_legs import write_bam
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, [(s, e) for s, e, _ in primers], 0)
    batch = synth.make_amplicon_batch(g, amps, 9000, seed=79)
    bam = str(tmp_path / "in.bam")
    write_bam(bam, batch, int(g.size))
    out = tmp_path / "ok"; out.mkdir()
    mp.spawn(_worker_file_partition, args=(world, _free_port(), bam, str(out), -1), nprocs=world, join=True)
    ref = oracle.process(batch, g.size, mn, mx, mpl, 20, 4)
    tally = np.zeros(g.size, np.int64); np.add.at(tally, ref.events["ref_pos"], 1)
    want_runs = Counter(event_strings(batch, ref.events))
    rows = 0
    for rank in range(world):
        got = np.load(str(out / ("r%d.npz" % rank)), allow_pickle=True)
        table = got["table"].view(np.uint32)
        assert np.array_equal(table[:g.size * 6].reshape(g.size, 6), ref.counts)
        assert np.array_equal(table[g.size * 6:], tally.astype(np.uint32))
        rel = set(int(p) for p in np.nonzero(tally)[0][::53])
        assert [tuple(x) for x in got["tallies"]] == sorted((p, s_, c) for (p, s_), c in want_runs.items() if p in rel)
        assert int(got["n_pieces"]) >= 2               # the share really was walked piece by piece
        rows += int(got["n_rows"])
    assert rows == batch.n                             # every record belongs to exactly one rank
    # the joined trimmed BAM of the ranks inflates to the bytes one writer makes of the whole file
    import gzip as gz
    whole = bam_native.BamFile(bam)
    wb, _ = whole.decode(0, whole.n_records, copy=True)
    keep = (ref.trim.ref_len >= 30) & ((ref.trim.trim_flags & 3) != 0)
    one = str(tmp_path / "one.bam")
    w1 = bam_native.BamWriter(one, whole.header_text, whole, level=1)
    w1.write_rows(None, wb.src_index, keep, ref.trim.new_pos, ref.trim.new_ncig, wb.cig_off[:-1] + np.uint64(3) * np.arange(wb.n, dtype=np.uint64), ref.trim.new_cig)
    w1.close(); whole.close()
    joined = open(str(out / "trim.bam"), "rb").read()
    assert joined.endswith(bam_native.BGZF_EOF) and gz.decompress(joined) == gz.decompress(open(one, "rb").read())
    shares = eval((out / "shares.txt").read_text())
    assert sum(r_ for r_, _ in shares) == batch.n and sum(b_ for _, b_ in shares) == int(batch.lseq.sum())
    # one rank fails in front of the collective: every rank learns of it instead of waiting in the all-reduce
    bad = tmp_path / "bad"; bad.mkdir()
    mp.spawn(_worker_file_partition, args=(world, _free_port(), bam, str(bad), world - 1), nprocs=world, join=True)
    for rank in range(world):
        msg = (bad / ("r%d.txt" % rank)).read_text()
        assert "rank %d: ValueError: made-up failure" % (world - 1) in msg
