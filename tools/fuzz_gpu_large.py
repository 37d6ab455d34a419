"""Larger randomised batches (hundreds of thousands of reads, many blocks, partly unsorted) against the oracle
(development aid; needs a GPU).  usage: fuzz_gpu_large.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amplipy_amd import synth
from amplipy_amd.batch import ReadBatch
from oracle import oracle
from tests.gpu_util import GpuRunner, assert_same
from tests.test_gpu_parity import _long_read_segments

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = [(s, e) for s, e, _ in primers]
runners = {v: GpuRunner(variant=v) for v in (2, 3, 4, 5)}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    off = int(rng.integers(0, 4)); mq = int(rng.choice([13, 20, 30])); w = int(rng.choice([3, 4, 5]))
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, off)
    pool = synth.make_mixed_segments(g, amps, 12000, seed=seed) + _long_read_segments(rng, 1500, int(g.size), 400, 30)
    pool += [s for s in synth.random_segments(rng, 3000, g.size, pr, domain_errors=False)]
    b0 = ReadBatch.from_segments(pool)
    a0 = oracle.process(b0, g.size, mn, mx, mpl, mq, w)
    pool = [pool[i] for i in np.nonzero(a0.trim.status == 0)[0]]
    rep = int(rng.integers(8, 25))
    segs = pool * rep
    if seed % 2:
        segs.sort(key=lambda s: s.reference_start)
    else:
        order = np.argsort(np.array([s.reference_start for s in segs]) + rng.integers(-200, 200, len(segs)))   # nearly sorted
        segs = [segs[i] for i in order]
    b = ReadBatch.from_segments(segs)
    t = time.time(); a = oracle.process(b, g.size, mn, mx, mpl, mq, w); to = time.time() - t
    for v, r in runners.items():
        d = r.process(b, g.size, mn, mx, mpl, mq, w)
        assert_same(a, d, b)
    print("seed %d: %d reads (x%d), mq %d w %d off %d: every kernel variant equals the oracle (oracle %.1fs)" % (seed, b.n, rep, mq, w, off, to), flush=True)
