"""Whole-command rates of the BAM path for several piece sizes (development aid; needs a GPU).
usage: e2e_sweep.py [piece KiB ...]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import amplipy, bam_native, synth
from tools.e2e_legs import write_bam

sizes = [int(x) for x in sys.argv[1:]] or [16384, 4096, 2048, 1024, 512]
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
hb = synth.make_amplicon_batch(g, amps, 150000, seed=4)
tmp = tempfile.mkdtemp(prefix="amp_sweep_")
seed = os.path.join(tmp, "seed.bam"); inp = os.path.join(tmp, "in.bam")
write_bam(seed, hb, G)
sf = bam_native.BamFile(seed); sb, _ = sf.decode(0, sf.n_records, copy=True)
w = bam_native.BamWriter(inp, sf.header_text, sf, level=6)
idx = np.repeat(np.arange(sb.n, dtype=np.int64), 10)
w.write_rows(None, sb.src_index[idx], np.ones(idx.size, np.uint8), sb.pos[idx], np.diff(sb.cig_off.astype(np.int64)).astype(np.uint32)[idx], sb.cig_off[:-1][idx], sb.cig)
w.close(); sf.close()
nb = idx.size
open(os.path.join(tmp, "ref.fas"), "w").write(">SYN_REF\n" + synth.genome_string(g) + "\n")
open(os.path.join(tmp, "p.bed"), "w").write("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e, i) for i, (s, e, _) in enumerate(primers)))
log = sys.stderr
for kib in sizes:
    os.environ["AMPLIPY_PART_BYTES"] = str(kib << 10)
    sys.stderr = open(os.devnull, "w")
    try:
        ta = tv = 1e9
        for it in range(3):
            o = {k: os.path.join(tmp, "%s_%d_%d.%s" % (k, kib, it, e)) for k, e in (("t", "bam"), ("v", "vcf"), ("c", "fas"))}
            t0 = time.perf_counter()
            amplipy.main(["aio", "-i", inp, "-p", os.path.join(tmp, "p.bed"), "-r", os.path.join(tmp, "ref.fas"), "-ot", o["t"], "-ov", o["v"], "-oc", o["c"]])
            ta = min(ta, time.perf_counter() - t0)
            t0 = time.perf_counter()
            amplipy.main(["variants", "-i", inp, "-r", os.path.join(tmp, "ref.fas"), "-o", os.path.join(tmp, "vv_%d_%d.vcf" % (kib, it))])
            tv = min(tv, time.perf_counter() - t0)
    finally:
        sys.stderr.close(); sys.stderr = log
    print("pieces of %5d KiB (%d pieces): aio %.1f ms = %.2f M reads/s, variants %.1f ms = %.2f M reads/s"
          % (kib, amplipy.native_parts(inp)[0], ta * 1e3, nb / ta / 1e6, tv * 1e3, nb / tv / 1e6), flush=True)
