"""cProfile of the `variants` / `aio` commands on the 1.5 M-read BAM of bench.py's e2e legs (development aid; needs a GPU).
usage: prof_variants.py [variants|aio]"""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import amplipy, bam_native, synth, synth_torch
from tools import e2e_legs

cmd = sys.argv[1] if len(sys.argv) > 1 else "variants"
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
b = synth_torch.make_amplicon_batch_device(g, amps, 150000, 1000, "cuda:0")
tmp = tempfile.mkdtemp(prefix="amp_prof_")
seed = os.path.join(tmp, "seed.bam"); inp = os.path.join(tmp, "in.bam")
e2e_legs.write_bam(seed, b.to_host(0, 150000), G)
sf = bam_native.BamFile(seed); sb, _ = sf.decode(0, sf.n_records, copy=True)
w = bam_native.BamWriter(inp, sf.header_text, sf, level=6)
idx = np.repeat(np.arange(sb.n, dtype=np.int64), 10)
w.write_rows(None, sb.src_index[idx], np.ones(idx.size, np.uint8), sb.pos[idx], np.diff(sb.cig_off.astype(np.int64)).astype(np.uint32)[idx], sb.cig_off[:-1][idx], sb.cig)
w.close(); sf.close()
open(os.path.join(tmp, "ref.fas"), "w").write(">SYN_REF\n" + synth.genome_string(g) + "\n")
open(os.path.join(tmp, "p.bed"), "w").write("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e, i) for i, (s, e, _) in enumerate(primers)))
def run(k):
    if cmd == "aio":
        amplipy.main(["aio", "-i", inp, "-p", os.path.join(tmp, "p.bed"), "-r", os.path.join(tmp, "ref.fas"), "-ot", os.path.join(tmp, "t%d.bam" % k), "-ov", os.path.join(tmp, "v%d.vcf" % k), "-oc", os.path.join(tmp, "c%d.fas" % k)])
    else:
        amplipy.main(["variants", "-i", inp, "-r", os.path.join(tmp, "ref.fas"), "-o", os.path.join(tmp, "vv%d.vcf" % k)])
err = sys.stderr; sys.stderr = open(os.devnull, "w")
run(0)
t0 = time.perf_counter(); run(1); t1 = time.perf_counter() - t0
pr = cProfile.Profile(); pr.enable(); run(2); pr.disable()
sys.stderr = err
print("%s: %.1f ms for 1.5 M reads -> %.2f M reads/s" % (cmd, t1 * 1e3, 1.5 / t1))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
