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
    t0 = time.perf_counter(); e.process_device(rd, 0, None); T["process_launch"] = T.get("process_launch", 0.0) + time.perf_counter() - t0; tick("process", t0)
    t0 = time.perf_counter(); k = e.last_kernel_ms(); tick("last_kernel_ms", t0)
    t0 = time.perf_counter(); res = calling.call(e, ref_seq, cp, None); tick("calling.call (compact)", t0)
    t0 = time.perf_counter(); s = res.consensus_string("N"); tick("consensus_string", t0)
for k, v in T.items(): print("%-36s %.3f ms" % (k, v / 20 * 1e3))
import ctypes as C
G = e.ref_len
T.clear()
for it in range(20):
    e.reset(); e.process_device(rd, 0, None); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = e.call_compact(cp); T["lib.call_compact"] = T.get("lib.call_compact", 0) + time.perf_counter() - t0
    t0 = time.perf_counter(); res = calling.result_from_compact(ref_seq, cp, *r[:2]); T["result_from_compact"] = T.get("result_from_compact", 0) + time.perf_counter() - t0
for k, v in T.items(): print("%-36s %.3f ms" % (k, v / 20 * 1e3))
