"""Config 5 (mixed 75-300 bp, soft clips, indel-heavy) throughput check (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from amplipy_amd import lib, synth
from amplipy_amd.batch import ReadBatch
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = [(s, e) for s, e, _ in primers]
pool = synth.make_mixed_segments(g, amps, 40000, seed=3)
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 25
segs = sorted(pool * rep, key=lambda s: s.reference_start)
t = time.time(); b = ReadBatch.from_segments(segs); print("pack %.1fs n=%d bases=%d" % (time.time() - t, b.n, b.total_bases()))
mn, mx, mpl = lib.find_overlapping_primers(g.size, pr, 0)
e = lib.Engine(g.size); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
for it in range(3):
    e.reset(); e.process(b, want_trim=False); tot, scan = e.last_kernel_ms()
    print("iter %d: kernels %.3f ms (tile %.3f) -> %.1f Mreads/s, %.2f Gbases/s; deferred=%d (heavy %d)" % (it, tot, scan, b.n / tot / 1e3, b.total_bases() / tot / 1e6, e.debug_counters()[3], e.debug_counters()[0]))
base = ReadBatch.from_segments(pool)
e.reset(); e.process(base, want_trim=False); c1 = e.counts()
e.reset(); e.process(b, want_trim=False); cN = e.counts()
assert np.array_equal(cN, c1 * np.uint32(rep)), "replication property violated"
print("replication property ok (x%d)" % rep)
