"""Nanopore-like amplicon reads (hundreds of bases, tens of CIGAR ops): every read takes the second pass
(development aid; needs a GPU).  usage: time_longreads.py [replication]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import lib
from amplipy_amd.batch import ReadBatch
from amplipy_amd.segment import Segment

G = 29903
rng = np.random.default_rng(7)
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 10
amps = [(s, s + 400) for s in range(30, G - 450, 300)]
primers = sorted([(a, a + 24) for a, b in amps] + [(b - 24, b) for a, b in amps])
pool = []
for _ in range(20000):
    a, b = amps[int(rng.integers(0, len(amps)))]
    ops, q, span = [], 0, 0
    while span < b - a - 10:
        m = int(rng.geometric(0.06)); m = min(m, b - a - span)
        ops.append((0, m)); q += m; span += m
        if span >= b - a - 10: break
        if rng.random() < 0.5:
            k = int(rng.integers(1, 4)); ops.append((1, k)); q += k
        else:
            k = int(rng.integers(1, 4)); ops.append((2, k)); span += k
    if ops[-1][0] != 0: ops.append((0, 5)); q += 5
    seq = "".join(rng.choice(list("ACGT"), q)); qual = rng.choice([30, 20, 12, 7], q, p=[0.5, 0.3, 0.15, 0.05]).tolist()
    pool.append(Segment(flag=int(rng.choice([0, 16])), reference_start=a + int(rng.integers(0, 4)), cigar=ops, template_length=0,
                        query_sequence=seq, query_qualities=qual))
segs = sorted(pool * rep, key=lambda s: s.reference_start)
b = ReadBatch.from_segments(segs)
nops = np.diff(b.cig_off)
print("n=%d reads, mean length %.0f, mean ops %.1f (max %d), bases %.1f M" % (b.n, b.lseq.mean(), nops.mean(), nops.max(), b.total_bases() / 1e6))
mn, mx, mpl = lib.find_overlapping_primers(G, primers, 0)
e = lib.Engine(G); e.set_primers(mn, mx, mpl); e.set_params(10, 4, True, True)
for it in range(3):
    e.reset(); e.process(b, want_trim=False); tot, scan = e.last_kernel_ms()
    print("iter %d: kernels %.3f ms (tile %.3f) -> %.1f M reads/s, %.2f G bases/s; errors %d" % (it, tot, scan, b.n / tot / 1e3, b.total_bases() / tot / 1e6, e.error_reads()))
for name, trim, count in (("trim only", True, False), ("count only", False, True)):
    e.set_params(10, 4, trim, count)
    for it in range(2):
        e.reset(); e.process(b, want_trim=False); tot, scan = e.last_kernel_ms()
    print("%s: kernels %.3f ms" % (name, tot))
e.set_params(10, 4, True, True)
base = ReadBatch.from_segments(pool)
e.reset(); e.process(base, want_trim=False); c1 = e.counts()
e.reset(); e.process(b, want_trim=False); cN = e.counts()
assert np.array_equal(cN, c1 * np.uint32(rep)), "replication property violated"
from oracle import oracle
a = oracle.process(base, G, mn, mx, mpl, 10, 4)
assert np.array_equal(a.counts, c1) and not a.trim.status.any(), "differs from the oracle"
print("replication property ok (x%d); base pool equals the oracle" % rep)
