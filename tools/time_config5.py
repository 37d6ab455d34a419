"""BASELINE config 5 at its full size: mixed 75-300 bp reads with long soft clips and indel-heavy CIGARs at 50k x depth
(about 8.0 M reads), one launch, checked against the C oracle (development aid; needs a GPU).
The batch is the 40,000-read pool of synth.make_mixed_segments gathered 200 times with numpy, every copy shifted by 0..7
positions and the whole sorted again, so packing takes seconds.
usage: time_config5.py [replication] [--no-check]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import lib, synth
from amplipy_amd.batch import ReadBatch


rep = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = [(s, e) for s, e, _ in primers]
t = time.time()
b = synth.make_config5_batch(g, amps, rep)
assert np.all(np.diff(b.pos.astype(np.int64)) >= -7)
nops = np.diff(b.cig_off.astype(np.int64))
print("packed %d reads in %.1f s: %.1f M bases, mean length %.0f, mean CIGAR ops %.1f (max %d), %.1f %% with an indel or more than one op"
      % (b.n, time.time() - t, b.total_bases() / 1e6, b.lseq.mean(), nops.mean(), nops.max(), 100.0 * (nops > 1).mean()))
mn, mx, mpl = lib.find_overlapping_primers(g.size, pr, 0)
e = lib.Engine(g.size); e.set_kernel_variant(int(os.environ.get("AMP_VARIANT", "0")) or 4) if os.environ.get("AMP_VARIANT") else None; e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.reserve_events(b.n // 2); e.set_timing(True)
for it in range(3):
    e.reset(); res = e.process(b); tot, scan = e.last_kernel_ms()
    dc = e.debug_counters()
    print("launch %d: all kernels %.3f ms (fast kernel %.3f) = %.3f ms per 1 M reads -> %.1f M reads/s, %.1f G bases/s; general-pass reads %d, of which %d left to the second pass"
          % (it, tot, scan, tot / (b.n / 1e6), b.n / tot / 1e3, b.total_bases() / tot / 1e6, int(dc[7]), int(dc[3])))
if os.environ.get("AMP_F7_STAMPS"):
    dc = e.debug_counters(); turns = max(int(dc[6]), 1)
    names = ["count pass tail + hand-over (to the top of the next turn)", "WAIT at the top", "next tile's shapes, bases issued, window, primer clips", "WAIT in the middle",
             "", "quality pass", "quality clip, extras' qualities", "count pass"]
    # stamps: 0 top-before-wait, 1 after, 4 before quality pass, 5 after, 2 before mid wait, 3 after, 6 before count pass, 7 after
    order = [1, 4, 5, 2, 3, 6, 7, 0]
    label = {1: "WAIT at the top of the turn", 4: "shapes of the next tile, bases issued, window, primer clips", 5: "quality pass (piece loop)", 2: "quality clip, qualities of the extras",
             3: "WAIT in the middle (the tile's bases)", 6: "results, next tile's requests, pad patch, indel extras", 7: "count pass (piece loop)", 0: "careful loop, hand-over, renames"}
    tot = float(sum(int(dc[8 + k]) for k in range(8)))
    print("k_fast7 stamps (last launch): %d turns, %.0f shader cycles per turn" % (turns, tot / turns))
    for k in order: print("   %-62s %8.0f  %5.1f %%" % (label[k], int(dc[8 + k]) / turns, 100.0 * int(dc[8 + k]) / tot))
if "--no-check" not in sys.argv:
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    counts = e.counts()
    cores = 16
    cuts = [b.n * k // cores for k in range(cores + 1)]
    def shard(k):
        sb = b.slice(cuts[k], cuts[k + 1])
        r = oracle.process(sb, g.size, mn, mx, mpl, 20, 4, read_base=cuts[k])
        assert np.array_equal(r.trim.new_pos, res.new_pos[cuts[k]:cuts[k + 1]]) and np.array_equal(r.trim.new_ncig, res.new_ncig[cuts[k]:cuts[k + 1]])
        assert np.array_equal(r.trim.status, res.status[cuts[k]:cuts[k + 1]])
        return r.counts, r.events.size
    t = time.time()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(shard, range(cores)))
    ref = sum(p[0].astype(np.uint64) for p in parts).astype(np.uint32)
    assert np.array_equal(ref, counts), "count table differs from the oracle"
    assert sum(p[1] for p in parts) == e.events().size, "number of insertion events differs from the oracle"
    print("check ok: count table, trimmed positions, op counts, statuses and the number of insertion events equal the C oracle's (%.1f s on %d threads)"
          % (time.time() - t, cores))
