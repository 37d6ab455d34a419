"""Where does a bench step spend its host time? (development aid)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from amplipy_amd import abi, calling, lib, synth, synth_torch
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = g.size
ref_seq = synth.genome_string(g)
n = synth.reads_for_depth(10000)
b = synth_torch.make_amplicon_batch_device(g, amps, n, 1000, "cuda:0"); torch.cuda.synchronize()
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
e = lib.Engine(G); e.set_stream(torch.cuda.current_stream().cuda_stream)
table = torch.zeros(G * 7, dtype=torch.int32, device="cuda:0"); e.bind_counts(table.data_ptr())
e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.set_reference(ref_seq); e.reserve_events(1 << 20)
rd = b.struct(); cp = calling.call_params(10, 0.0, 1, 0.03, True, True)
T = {}
def tick(k, t0):
    torch.cuda.synchronize(); T[k] = T.get(k, 0.0) + time.perf_counter() - t0
for it in range(25):
    if it == 5: T.clear()
    t0 = time.perf_counter(); e.reset(); tick("reset", t0)
    t0 = time.perf_counter(); e.process_device(rd, 0, None); tick("process", t0)
    t0 = time.perf_counter(); pcs = e.call_positions(cp); tick("call_positions", t0)
    t0 = time.perf_counter(); c = e.counts(); tick("counts_d2h", t0)
    t0 = time.perf_counter(); res = calling.call(e, ref_seq, cp, None, positions=pcs); tick("calling.call(total incl counts)", t0)
    t0 = time.perf_counter(); s = res.consensus_string("N"); tick("consensus_string", t0)
for k, v in T.items(): print("%-36s %.3f ms" % (k, v / 20 * 1e3))
