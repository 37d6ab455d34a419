"""Where the second pass spends its time on config 5: trim only / count only / both (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import lib, synth
from amplipy_amd.batch import ReadBatch
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = [(s, e) for s, e, _ in primers]
pool = synth.make_mixed_segments(g, amps, 40000, seed=3)
segs = sorted(pool * 5, key=lambda s: s.reference_start)
b = ReadBatch.from_segments(segs)
mn, mx, mpl = lib.find_overlapping_primers(g.size, pr, 0)
e = lib.Engine(g.size); e.set_primers(mn, mx, mpl)
for name, trim, count, mq in (("both", True, True, 20), ("trim only", True, False, 20), ("count only", False, True, 20),
                              ("both, mq=0", True, True, 0)):
    e.set_params(mq, 4, trim, count)
    for it in range(2):
        e.reset(); e.process(b, want_trim=False); tot, scan = e.last_kernel_ms()
    print("%-12s kernels %.3f ms (tile %.3f); deferred=%d" % (name, tot, scan, e.debug_counters()[3]))
