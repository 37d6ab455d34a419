"""Wall time of the command line on a synthetic BAM (development aid; needs a GPU).  usage: time_cli.py [n_reads]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import amplipy, bamio, synth
from amplipy_amd.batch import SEQ_NT16, unpack_nibbles

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
b = synth.make_amplicon_batch(g, amps, n, seed=4)
tmp = tempfile.mkdtemp()
inp = os.path.join(tmp, "in.bam")
open(os.path.join(tmp, "ref.fas"), "w").write(">SYN_REF\n" + synth.genome_string(g) + "\n")
open(os.path.join(tmp, "p.bed"), "w").write("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e, i) for i, (s, e, _) in enumerate(primers)))
hdr = bamio.Header("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:SYN_REF\tLN:%d\n@PG\tID:sim\tPN:sim\n" % G, [("SYN_REF", G)])
t = time.time()
w = bamio.AlignmentWriter(inp, "wb", hdr)
lut = np.frombuffer(SEQ_NT16.encode(), np.uint8)
for i in range(b.n):
    o = int(b.seq_off[i]); L = int(b.lseq[i])
    seq = lut[unpack_nibbles(b.seq[o // 2:(o + L + 1) // 2], L)].tobytes().decode()
    a, c = int(b.cig_off[i]), int(b.cig_off[i + 1])
    w.write(bamio.Rec("r%d" % i, int(b.flag[i]), 0, int(b.pos[i]), 60, [(int(v) & 15, int(v) >> 4) for v in b.cig[a:c]], 0, int(b.pos[i]),
                      int(b.tlen[i]), seq, bytes(b.qual[o:o + L])))
w.close()
print("wrote %d-read BAM in %.1fs" % (b.n, time.time() - t), file=sys.stderr)
for it, cmd in enumerate(("aio", "aio", "variants")):
    outs = {k: os.path.join(tmp, "%s%d.%s" % (k, it, ext)) for k, ext in (("t", "bam"), ("v", "vcf"), ("c", "fas"))}
    if cmd == "aio":
        argv = ["aio", "-i", inp, "-p", os.path.join(tmp, "p.bed"), "-r", os.path.join(tmp, "ref.fas"), "-ot", outs["t"], "-ov", outs["v"], "-oc", outs["c"]]
    else:
        argv = ["variants", "-i", os.path.join(tmp, "t0.bam"), "-r", os.path.join(tmp, "ref.fas"), "-o", outs["v"]]
    t0 = time.perf_counter()
    if it == 1 and os.environ.get("PROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); amplipy.main(argv); pr.disable()
        pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative").print_stats(22)
    else:
        amplipy.main(argv)
    dt = time.perf_counter() - t0
    print("%s: %.3fs -> %.2f M reads/s (whole command)" % (cmd, dt, b.n / dt / 1e6))
