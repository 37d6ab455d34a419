"""Where the 30-60 ms behind the read loop of a `variants` run go (development aid; needs a GPU).  usage: cc_time2.py in.bam"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amplipy_amd import lib, synth, calling, bam_native, synth_torch
from tools import e2e_legs
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
tmp = tempfile.mkdtemp(); seed = os.path.join(tmp, "s.bam"); inp = os.path.join(tmp, "in.bam")
b = synth_torch.make_amplicon_batch_device(g, amps, 150000, 1000, "cuda:0")
e2e_legs.write_bam(seed, b.to_host(0, 150000), G)
sf = bam_native.BamFile(seed); sb, _ = sf.decode(0, sf.n_records, copy=True)
w = bam_native.BamWriter(inp, sf.header_text, sf, level=6)
idx = np.repeat(np.arange(sb.n, dtype=np.int64), 10)
w.write_rows(None, sb.src_index[idx], np.ones(idx.size, np.uint8), sb.pos[idx], np.diff(sb.cig_off.astype(np.int64)).astype(np.uint32)[idx], sb.cig_off[:-1][idx], sb.cig)
w.close(); sf.close()
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
cp = calling.call_params(10, 0.0, 1, 0.03, True, True)
for mode in ("plain", "plain", "sleep", "sync", "noclose", "nostore", "plain"):
    T = time.perf_counter
    t0 = T(); eng = lib.Engine(G); eng.set_primers(mn, mx, mpl); eng.set_params(20, 4, False, True); eng.set_reference(synth.genome_string(g))
    t1 = T(); src = bam_native.BamFile(inp); t2 = T()
    td = tp = ts = 0.0
    for first in range(0, src.n_records, 250000):
        a = T(); batch, _ = src.decode(first, min(250000, src.n_records - first)); c = T(); eng.process(batch, read_base=first); d = T()
        if mode != "nostore": eng.aggregate_events(None, first, drain=True)
        e = T(); td += c - a; tp += d - c; ts += e - d
    t3 = T()
    if mode != "noclose": src.close()
    t4 = T()
    if mode == "sleep": time.sleep(0.1)
    if mode == "sync": eng.sync()
    t5 = T(); eng.call_compact(cp); t6 = T(); eng.close(); t7 = T()
    if mode == "noclose": src.close()
    print("%-8s engine %.1f open %.1f decode %.1f process %.1f events %.1f bamclose %.1f [%s %.1f] call_compact %.1f engclose %.1f" %
          (mode, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*td, 1e3*tp, 1e3*ts, 1e3*(t4-t3), mode, 1e3*(t5-t4), 1e3*(t6-t5), 1e3*(t7-t6)))
