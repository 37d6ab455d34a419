"""In-kernel phase stamps of k_fast (development aid; needs a -DAMP_DEV build of the library, AMPLIHIP_LIB=...):
shader cycles per tile and wave in each phase of the tile loop, and the shader clock the kernel ran at.
usage: AMP_STAMPS=1 AMPLIHIP_LIB=<dev build> python tools/stamp_phases.py [--depth D] [--waves W]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import abi, lib, synth, synth_torch
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=10000); ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--waves", type=int, default=8, help="waves per block of the build (F_WAVES)")
a = ap.parse_args()
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = g.size
n = synth.reads_for_depth(a.depth)
b = synth_torch.make_amplicon_batch_device(g, amps, n, 1000, "cuda:0"); torch.cuda.synchronize()
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
e = lib.Engine(G); e.set_kernel_variant(4)
e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.reserve_events(max(1 << 20, n // 4))
out = {k: torch.zeros(sz, dtype=dt, device="cuda:0") for k, sz, dt in
       (("new_pos", n, torch.int32), ("new_ncig", n, torch.int32), ("new_cig", b.n_cig + 3 * n, torch.int32),
        ("ref_len", n, torch.int32), ("trim_flags", n, torch.uint8), ("status", n, torch.uint8))}
dev_out = abi.AmpTrimOut(*[out[k].data_ptr() for k in ("new_pos", "new_ncig", "new_cig", "ref_len", "trim_flags", "status")])
rd = b.struct()
ms = []
for it in range(a.iters):
    e.reset(); e.process_device(rd, 0, dev_out); e.sync(); ms.append(e.last_kernel_ms()[0])
dc = [int(x) for x in e.debug_counters()]
nt = (n + 63) // 64
nw = a.waves * 256
names = ("top wait", "rows+clips+issue", "scan", "qclip+results", "count", "careful+handover")
ph = [dc[8 + k] / nt for k in range(1, 7)]
print("depth %d waves/block %d: pass ms %s" % (a.depth, a.waves, ["%.3f" % m for m in ms]))
print("shader clock %.0f MHz (kernel cycles / wall per wave); kernel %.0f cycles per wave, tile loop %.0f" %
      (100.0 * dc[8] / max(dc[15], 1), dc[8] / nw, sum(dc[9:15]) / nw))
print("cycles per tile and wave: " + " | ".join("%s %.0f" % (nm, v) for nm, v in zip(names, ph)) + " | total %.0f" % sum(ph))
