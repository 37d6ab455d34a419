"""Ad-hoc timing of the hot path on a host-generated batch (development aid, not the bench)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from amplipy_amd import lib, synth

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = [(s, e) for s, e, _ in primers]
t = time.time(); b = synth.make_amplicon_batch(g, amps, synth.reads_for_depth(depth), seed=1); print("gen %.1fs n=%d" % (time.time() - t, b.n))
mn, mx, mpl = lib.find_overlapping_primers(g.size, pr, 0)
e = lib.Engine(g.size); e.set_kernel_variant(variant); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
for it in range(4):
    e.reset(); t = time.time(); e.process(b, want_trim=False); dt = time.time() - t
    tot, scan = e.last_kernel_ms()
    print("iter %d: wall %.2f ms, kernels %.3f ms, scan %.3f ms -> %.2f Mreads/s (kernel)" % (it, dt * 1e3, tot, scan, b.n / tot / 1e3))
dc = e.debug_counters()
if dc[13]:
    names = ["P1", "P2", "P3", "P4", "tail"]
    tot_c = float(sum(dc[8:13]))
    print("defer reasons: irregular=%d segcap=%d cerr=%d chunk_err=%d" % (dc[6], dc[7], dc[14], dc[15]))
    print("stamps: tiles=%d deferred=%d " % (dc[13], dc[3]) + " ".join("%s=%.0f cyc/tile (%.0f%%)" % (n_, dc[8 + k] / dc[13], 100.0 * dc[8 + k] / tot_c) for k, n_ in enumerate(names)))
