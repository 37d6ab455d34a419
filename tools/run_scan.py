"""Minimal driver for profiling the scan kernels (development aid): builds the bench workload in HBM and runs
amp_process_batch_device a few times.  usage: run_scan.py [--depth D] [--iters K] [--variant V] [--check]"""
import argparse, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import abi, lib, synth, synth_torch
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=10000); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--variant", type=int, default=4); ap.add_argument("--check", action="store_true")
a = ap.parse_args()
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = g.size
n = synth.reads_for_depth(a.depth)
b = synth_torch.make_amplicon_batch_device(g, amps, n, 1000, "cuda:0"); torch.cuda.synchronize()
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
e = lib.Engine(G); e.set_kernel_variant(a.variant)
e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.reserve_events(max(1 << 20, n // 4))
out = {k: torch.zeros(sz, dtype=dt, device="cuda:0") for k, sz, dt in
       (("new_pos", n, torch.int32), ("new_ncig", n, torch.int32), ("new_cig", b.n_cig + 3 * n, torch.int32),
        ("ref_len", n, torch.int32), ("trim_flags", n, torch.uint8), ("status", n, torch.uint8))}
dev_out = abi.AmpTrimOut(*[out[k].data_ptr() for k in ("new_pos", "new_ncig", "new_cig", "ref_len", "trim_flags", "status")])
rd = b.struct()
ms = []
for it in range(a.iters):
    e.reset(); e.process_device(rd, 0, dev_out); e.sync(); ms.append(e.last_kernel_ms())
print("variant %d depth %d reads %d: total/scan ms per launch:" % (a.variant, a.depth, n), ["%.3f/%.3f" % m for m in ms])
if os.environ.get("AMP_STAMPS"):
    dc = e.debug_counters()
    tot = float(sum(int(x) for x in dc[8:15])) or 1.0
    print("phase shares (header wait, bytes wait, staging, clips+scan, qclip+outputs, count, careful+handover):",
          ["%.1f%%" % (100.0 * int(x) / tot) for x in dc[8:15]], "cycles/tile-wave %.0f" % (tot / a.iters / ((n + 63) // 64)))
if a.check:
    from oracle import oracle
    hb = b.to_host()
    ref = oracle.process(hb, G, mn, mx, mpl, 20, 4)
    assert np.array_equal(e.counts(), ref.counts), "counts differ"
    assert np.array_equal(out["new_pos"].cpu().numpy(), ref.trim.new_pos)
    print("check ok")
