"""End-to-end timing of the BAM path (development aid): BAM file -> libampbam decode -> GPU trim + pileup
-> re-encoded trimmed BAM + calls.  Needs a GPU.  usage: time_bam_e2e.py [n_reads]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import bam_native, bamio, calling, lib, synth
from amplipy_amd.batch import SEQ_NT16, unpack_nibbles

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = int(g.size)
b = synth.make_amplicon_batch(g, amps, n, seed=4)
tmp = tempfile.mkdtemp()
inp = os.path.join(tmp, "in.bam"); out = os.path.join(tmp, "out.bam")
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
print("wrote %d-read BAM with the Python codec in %.1fs (%.1f MB)" % (b.n, time.time() - t, os.path.getsize(inp) / 1e6))
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
eng = lib.Engine(G); eng.set_primers(mn, mx, mpl); eng.set_params(20, 4, True, True); eng.set_reference(synth.genome_string(g))
cp = calling.call_params(10, 0.0, 1, 0.03, True, True)
for it in range(2):
    if os.path.exists(out): os.remove(out)
    T = {}
    t0 = time.perf_counter(); src = bam_native.BamFile(inp); T["open+inflate"] = time.perf_counter() - t0
    wr = bam_native.BamWriter(out, hdr.with_amplipy_pg("0.0.2", "x").text, src, level=1)
    eng.reset()
    t0 = time.perf_counter(); batch, _ = src.decode(0, src.n_records); T["decode"] = time.perf_counter() - t0
    t0 = time.perf_counter(); res = eng.process(batch); T["gpu (H2D + kernels + D2H)"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    keep = (res.ref_len >= 30) & ((res.trim_flags & 3) != 0)
    slot = batch.cig_off[:-1] + np.uint64(3) * np.arange(batch.n, dtype=np.uint64)
    wr.write_rows(None, batch.src_index, keep, res.new_pos, res.new_ncig, slot, res.new_cig); wr.close(); T["re-encode + deflate(level 1) + write"] = time.perf_counter() - t0
    t0 = time.perf_counter(); r = calling.call(eng, synth.genome_string(g), cp, None); T["call"] = time.perf_counter() - t0
    src.close()
    tot = sum(T.values())
    print("iter %d: total %.3fs -> %.2f M reads/s end to end; " % (it, tot, b.n / tot / 1e6) + ", ".join("%s %.3fs" % kv for kv in T.items()))
