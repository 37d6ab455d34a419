"""Find the reads on which a kernel variant and the oracle disagree (development aid): the batch of
tests/test_gpu_parity.py::test_regular_cigars_with_awkward_indels for a seed, bisected down to single reads."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd.batch import ReadBatch
from oracle import oracle
from tests.gpu_util import GpuRunner
from tests.test_gpu_parity import _awkward_regular_segments

seed, mq, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 6
G = 5000
rng = np.random.default_rng(seed)
primers = sorted((int(s), int(s) + int(rng.integers(18, 31))) for s in rng.integers(0, G - 40, 40))
mn, mx, mpl = oracle.find_overlapping_primers(G, primers, int(rng.integers(0, 3)))
segs = _awkward_regular_segments(rng, 6000, G)
segs.sort(key=lambda s: s.reference_start)
b = ReadBatch.from_segments(segs)
r = GpuRunner(variant=variant)
for do_trim in (True, False):
    a = oracle.process(b, G, mn, mx, mpl, mq, w, do_trim=do_trim)
    ok = np.nonzero(a.trim.status == 0)[0]
    good = [segs[i] for i in ok]
    def differs(ss):
        bb = ReadBatch.from_segments(ss)
        x = oracle.process(bb, G, mn, mx, mpl, mq, w, do_trim=do_trim); y = r.process(bb, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        return not np.array_equal(x.counts, y.counts), x, y
    d, x, y = differs(good)
    print("do_trim", do_trim, "reads", len(good), "differs", d)
    if not d:
        continue
    pos = np.nonzero((x.counts != y.counts).any(axis=1))[0]
    print("positions:", pos[:20], "oracle", x.counts[pos[:5]].tolist(), "device", y.counts[pos[:5]].tolist())
    # single reads
    found = 0
    for k, s in enumerate(good):
        dd, xx, yy = differs([s])
        if dd:
            pp = np.nonzero((xx.counts != yy.counts).any(axis=1))[0]
            print("read", k, "pos", s.reference_start, "flag", s.flag, "cigar", s.cigartuples, "len", s.query_length, "->", xx.trim.compact_cigars().tolist(), "newpos", xx.trim.new_pos.tolist(),
                  "diff at", pp[:8], xx.counts[pp[:4]].tolist(), yy.counts[pp[:4]].tolist())
            print("   seq", s.query_sequence); print("   qual", s.query_qualities)
            found += 1
            if found >= 4:
                break
    if not found:
        print("no single read differs: an interaction")
