import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from amplipy_amd import lib, synth, calling
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
from amplipy_amd import synth_torch
b = synth_torch.make_amplicon_batch_device(g, amps, int(sys.argv[1]) if len(sys.argv) > 1 else 250000, 1000, 'cuda:0').to_host()
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
cp = calling.call_params(10, 0.0, 1, 0.03, True, True)
for rnd in range(3):
    t0 = time.perf_counter(); eng = lib.Engine(G); t1 = time.perf_counter()
    eng.set_primers(mn, mx, mpl); eng.set_params(20, 4, True, True); eng.set_reference(synth.genome_string(g)); t2 = time.perf_counter()
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6): eng.process(b)
    t3 = time.perf_counter()
    eng.call_compact(cp); t4 = time.perf_counter()
    eng.call_compact(cp); t5 = time.perf_counter()
    eng.close(); t6 = time.perf_counter()
    print("engine %.1f ms, setup %.1f, process %.1f, call_compact first %.1f, second %.2f, close %.1f" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
