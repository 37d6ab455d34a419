import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import lib, synth
from amplipy_amd.batch import ReadBatch
from oracle import oracle
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = g.size
pr = [(s, e) for s, e, _ in primers]
mn, mx, mpl = lib.find_overlapping_primers(G, pr, 0)
b0 = synth.make_amplicon_batch(g, amps, 40, seed=7, indel_frac=0.0)
segs = b0.segments()
for reps in (1, 2, 8, 9, 17, 33, 64, 65, 130):
    b = ReadBatch.from_segments([segs[0]] * reps)
    ref = oracle.process(b, G, mn, mx, mpl, 20, 4)
    e = lib.Engine(G); e.set_kernel_variant(4); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
    res = e.process(b); c = e.counts()
    d = np.argwhere(c != ref.counts)
    print("reps=%d: %d differing cells, sum gpu %d ref %d" % (reps, len(d), c.sum(), ref.counts.sum()))
    for p, s in d[:12]:
        print("   pos %d (rel %d) sym %s gpu %d ref %d" % (p, p - res.new_pos[0], "ACGTN-"[s], c[p, s], ref.counts[p, s]))
    e.close()
