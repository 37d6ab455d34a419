"""Randomised GPU-vs-oracle comparison beyond the test-suite's fixed seeds (development aid; needs a GPU).
usage: fuzz_gpu.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amplipy_amd import synth
from amplipy_amd.batch import ReadBatch
from oracle import oracle
from tests.gpu_util import GpuRunner, assert_same
from tests.test_gpu_parity import _long_read_segments

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = [(s, e) for s, e, _ in primers]
runners = {v: GpuRunner(variant=v) for v in (1, 2, 3, 4, 5, 6, 7)}
t0 = time.time(); n_cmp = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    mq = int(rng.choice([0, 2, 13, 20, 30, 41])); w = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 30])); off = int(rng.integers(0, 8))
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, off)
    kind = seed % 4
    if kind == 0:
        segs = synth.random_segments(rng, 5000, g.size, pr)
    elif kind == 3:
        segs = synth.random_segments(rng, 5000, g.size, pr, max_len=304)        # (the length bins and lane pairs of the list-driven fast kernel)
    elif kind == 1:
        segs = synth.make_mixed_segments(g, amps, 5000, seed=seed)
    else:
        segs = _long_read_segments(rng, 1500, int(g.size), int(rng.choice([300, 1200, 5000])), int(rng.choice([8, 14, 18, 22, 40, 90])))
    if rng.random() < 0.5:
        segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, g.size, mn, mx, mpl, mq, w)
    ok = np.nonzero(a.trim.status == 0)[0]
    good = ReadBatch.from_segments([segs[i] for i in ok])
    for do_trim in (True, False):
        a2 = oracle.process(good, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
        bad = a2.trim.status != 0
        gb = good if not bad.any() else ReadBatch.from_segments([segs[ok[i]] for i in np.nonzero(~bad)[0]])
        if bad.any():
            a2 = oracle.process(gb, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
        for v, r in runners.items():
            d = r.process(b, g.size, mn, mx, mpl, mq, w)
            assert_same(a, d, b, check_counts=False)
            d2 = r.process(gb, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
            assert_same(a2, d2, gb)
            n_cmp += 2
    print("seed %d kind %d mq %d w %d off %d: %d reads (%d without errors) ok" % (seed, kind, mq, w, off, b.n, good.n), flush=True)
print("all equal: %d comparisons in %.0fs" % (n_cmp, time.time() - t0))
