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
    nb = max(int(dc[15]) // 8, 1)
    print("waves=%d: mean loop %.0f cyc, mean wait at final barrier %.0f cyc; blocks=%d mean %.0f cyc, slowest %d cyc (block %d), fastest %d cyc (block %d)" % (dc[15], dc[6] / max(dc[15], 1), dc[7] / max(dc[15], 1), nb, dc[4] / nb, int(dc[14]) >> 16, int(dc[14]) & 65535, 0xFFFFFFFFFFFF - (int(dc[5]) >> 16), int(dc[5]) & 65535))
    print("stamps: tiles=%d deferred=%d " % (dc[13], dc[3]) + " ".join("%s=%.0f cyc/tile (%.0f%%)" % (n_, dc[8 + k] / dc[13], 100.0 * dc[8 + k] / tot_c) for k, n_ in enumerate(names)))

if dc[13]:
    bl = e.debug_blocks().astype(np.int64)
    order = np.argsort(bl[:, 0])
    print("block stats [dur, rebases, p2 chunks, p4 chunks]: fastest", bl[order[:3]].tolist(), "median", bl[order[len(order) // 2]].tolist(), "slowest", bl[order[-5:]].tolist(), "ids", order[-5:].tolist())
    print("corr(dur, p2)=%.2f corr(dur, p4)=%.2f corr(dur, rebases)=%.2f" % tuple(np.corrcoef(bl[:-1, 0], bl[:-1, k])[0, 1] for k in (2, 3, 1)))
if dc[13]:
    d = bl[:-1, 0].astype(float)
    ids = np.arange(d.size)
    print("mean block cycles by (block %% 8):", [int(d[ids % 8 == k].mean()) for k in range(8)])
    print("by (block // 8) %% 4:", [int(d[(ids // 8) % 4 == k].mean()) for k in range(4)], " by block//32:", [int(d[ids // 32 == k].mean()) for k in range(8)])
    print("sorted durations (every 16th):", np.sort(d)[::16].astype(int).tolist())
