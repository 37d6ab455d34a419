"""Fast kernels on a batch of MIXED read lengths with simple CIGARs (adapter- or quality-trimmed 2 x 250 runs look like this): amplicon
reads of 100 / 150 / 200 / 250 bases, a quarter each, interleaved by position (development aid; needs a GPU).
usage: mixed_simple.py [reads per length]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import lib, synth
from amplipy_amd.batch import ReadBatch


def concat(bs):
    cig_off = [np.zeros(1, np.uint64)]; seq_off = [np.zeros(1, np.uint64)]
    c0 = s0 = 0
    for b in bs:
        cig_off.append(b.cig_off[1:] + np.uint64(c0)); seq_off.append(b.seq_off[1:] + np.uint64(s0))
        c0 += int(b.cig_off[-1]); s0 += int(b.seq_off[-1])
    cat = lambda k: np.concatenate([getattr(b, k) for b in bs])
    return ReadBatch(cat("pos"), cat("flag"), cat("tlen"), cat("lseq"), np.concatenate(cig_off), cat("cig"), np.concatenate(seq_off), cat("seq"), cat("qual"))


per = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
mn, mx, mpl = lib.find_overlapping_primers(g.size, [(s, e) for s, e, _ in primers], 0)
b = concat([synth.make_amplicon_batch(g, amps, per, seed=40 + k, read_len=L) for k, L in enumerate((100, 150, 200, 250))])
b = synth.gather_rows(b, np.argsort(b.pos, kind="stable"))
print("%d reads, mean length %.0f, mean CIGAR ops %.2f" % (b.n, b.lseq.mean(), b.cig.size / b.n))
ref = None
for v in (0, 5, 7, 2):
    e = lib.Engine(g.size); e.set_kernel_variant(v); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True); e.reserve_events(b.n // 2); e.set_timing(True)
    ms = []
    for it in range(4):
        e.reset(); res = e.process(b); ms.append(e.last_kernel_ms())
    c = e.counts()
    if ref is None: ref = (c, res.new_pos.copy(), res.new_ncig.copy())
    same = np.array_equal(c, ref[0]) and np.array_equal(res.new_pos, ref[1]) and np.array_equal(res.new_ncig, ref[2])
    print("variant %d (took %d): all kernels / fast kernel ms %s; general-pass reads %d; equal to the first variant's results: %s"
          % (v, e.last_kernel_variant(), ["%.3f/%.3f" % m for m in ms[1:]], int(e.debug_counters()[7]), same))
    e.close()
