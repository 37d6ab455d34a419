import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from amplipy_amd import lib, synth, synth_torch
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
n = synth.reads_for_depth(10000)
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
b = synth_torch.make_amplicon_batch_device(g, amps, n, 1000, "cuda:0"); torch.cuda.synchronize()
rd = b.struct()
ref = None
for v in (2, 3):
    e = lib.Engine(G); e.set_kernel_variant(v); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.reserve_events(1 << 22)
    ts = []
    for it in range(8):
        e.reset(); e.process_device(rd, 0, None); e.sync(); ts.append(e.last_kernel_ms())
    c = e.counts()
    if ref is None: ref = c
    print("variant %d: scan kernels %.4f ms, all kernels %.4f ms, counts equal to variant 2: %s" % (v, np.mean([t[1] for t in ts[3:]]), np.mean([t[0] for t in ts[3:]]), np.array_equal(c, ref)))
    e.close()
