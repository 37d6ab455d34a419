"""In-kernel phase stamps of the general pass (k_tile<LIST>) on a BASELINE config-5 batch (development aid; needs a GPU and a
-DAMP_DEV build of the library: the shipped one has no stamps).
usage: AMPLIHIP_PHASES=0x1FF AMPLIHIP_LIB=tools/micro/bin/libamplihip_dev.so python tools/stamp_tile.py [replication]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import lib, synth

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
b = synth.make_config5_batch(g, amps, rep)
mn, mx, mpl = lib.find_overlapping_primers(g.size, [(s, e) for s, e, _ in primers], 0)
e = lib.Engine(g.size); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.reserve_events(b.n // 2)
for it in range(3):
    e.reset(); e.process(b); tot, scan = e.last_kernel_ms()
    dc = [int(x) for x in e.debug_counters()]
print("%d reads, all kernels %.3f ms" % (b.n, tot))
waves, tiles = dc[15], dc[13]
names = ["P1 loads + primer clips", "P2 window scan (chunk lanes)", "P3 quality clip, outputs, segments", "P4 match bases (chunk lanes)", "indel walk + hand-over"]
tot_c = sum(dc[8:13])
print("waves %d, tiles %d (reads in the list / 64 = %.0f), tile loop %.0f cycles per wave (memtime units: 100 MHz), barrier wait %.0f" % (waves, tiles, dc[7] / 64.0, dc[6] / max(waves, 1), dc[7] / max(waves, 1)))
for k, nm in enumerate(names):
    print("  %-36s %8.1f ticks per tile  %5.1f %%" % (nm, dc[8 + k] / max(tiles, 1), 100.0 * dc[8 + k] / max(tot_c, 1)))
print("  total %.1f ticks of 10 ns per tile and wave" % (tot_c / max(tiles, 1)))
