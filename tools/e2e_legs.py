"""File-to-file legs of bench.py's `e2e` object (run after the timed region; needs a GPU).

  host_ptr_reads_per_s      amp_process_batch from host arrays: H2D staging + kernels + D2H of the per-read results
  bam_to_calls_reads_per_s  the `variants` command on a BAM file: inflate + decode + H2D + kernels + calling + VCF
  bam_to_bam_reads_per_s    the `aio` command: the same plus re-encoding and deflating the trimmed BAM, VCF, consensus
All three are whole-call wall times on a bounded sample of the bench workload (the BAM is written from the first
rows of the same synthetic batch with the package's own codec).
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def write_bam(path, hb, G):
    from amplipy_amd import bamio
    from amplipy_amd.batch import SEQ_NT16, unpack_nibbles
    hdr = bamio.Header("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:SYN_REF\tLN:%d\n@PG\tID:sim\tPN:sim\n" % G, [("SYN_REF", G)])
    w = bamio.AlignmentWriter(path, "wb", hdr)
    lut = np.frombuffer(SEQ_NT16.encode(), np.uint8)
    for i in range(hb.n):
        o = int(hb.seq_off[i]); L = int(hb.lseq[i])
        seq = lut[unpack_nibbles(hb.seq[o // 2:(o + L + 1) // 2], L)].tobytes().decode()
        a, c = int(hb.cig_off[i]), int(hb.cig_off[i + 1])
        w.write(bamio.Rec("r%d" % i, int(hb.flag[i]), 0, int(hb.pos[i]), 60, [(int(v) & 15, int(v) >> 4) for v in hb.cig[a:c]], 0,
                          int(hb.pos[i]), int(hb.tlen[i]), seq, bytes(hb.qual[o:o + L])))
    w.close()


def measure(batch, genome, primers, ref_seq, dev, n_host=1000000, n_bam=1500000):
    from amplipy_amd import amplipy, lib
    G = int(genome.size)
    out = {}
    mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
    # ---- host pointers ----
    nh = min(batch.n, n_host)
    hb = batch.to_host(0, nh)
    eng = lib.Engine(G)
    eng.set_primers(mn, mx, mpl); eng.set_params(20, 4, True, True)
    best = 1e9
    for _ in range(3):
        eng.reset()
        t0 = time.perf_counter(); eng.process(hb); best = min(best, time.perf_counter() - t0)
    eng.close()
    out["host_ptr_reads_per_s"] = round(nh / best, 1)
    out["host_ptr_sample"] = "%d reads, amp_process_batch (pageable host arrays in, per-read results out), best of 3" % nh
    # ---- BAM legs: a 1.5 M-read file of DISTINCT records: the first rows of the batch, written by libampbam from the packed
    # arrays (ampbam_write_batch; a 64-read file of the Python codec lends its header and reference dictionary), so that the
    # DEFLATE streams have the entropy of a real amplicon run (random qualities) and start-up does not dominate the commands ----
    nb = min(batch.n, n_bam)
    tmp = tempfile.mkdtemp(prefix="amp_e2e_")
    seed = os.path.join(tmp, "seed.bam")
    write_bam(seed, batch.to_host(0, 64), G)
    from amplipy_amd import bam_native
    inp = os.path.join(tmp, "in.bam")
    sf = bam_native.BamFile(seed)
    hbb = batch.to_host(0, nb)
    w = bam_native.BamWriter(inp, sf.header_text, sf, level=6)
    w.write_batch(hbb)
    w.close(); sf.close()
    raw_bytes = int(hbb.n * (36 + 10) + 4 * hbb.cig.size + hbb.lseq.astype(np.int64).sum() * 3 // 2)
    del hbb
    with open(os.path.join(tmp, "ref.fas"), "w") as f:
        f.write(">SYN_REF\n" + ref_seq + "\n")
    with open(os.path.join(tmp, "p.bed"), "w") as f:
        f.write("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e, i) for i, (s, e, _) in enumerate(primers)))
    # ---- the stages one after the other (what each costs on its own; the commands below overlap them) ----
    try:
        stages = {}
        t0 = time.perf_counter(); src = bam_native.BamFile(inp); stages["open_inflate_index_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); bb, _ = src.decode(0, src.n_records); stages["decode_to_packed_batch_ms"] = (time.perf_counter() - t0) * 1e3
        eng = lib.Engine(G); eng.set_primers(mn, mx, mpl); eng.set_params(20, 4, True, True)
        eng.process(bb); eng.reset()
        t0 = time.perf_counter(); res = eng.process(bb); stages["gpu_h2d_kernels_d2h_ms"] = (time.perf_counter() - t0) * 1e3
        stages["gpu_kernels_only_ms"] = eng.last_kernel_ms()[0]
        wr = bam_native.BamWriter(os.path.join(tmp, "stage.bam"), src.header_text, src, level=int(os.environ.get("AMPLIPY_BAM_LEVEL", "-1")))
        keep = (res.ref_len >= 30) & ((res.trim_flags & 3) != 0)
        slot = bb.cig_off[:-1] + np.uint64(3) * np.arange(bb.n, dtype=np.uint64)
        t0 = time.perf_counter(); wr.write_rows(None, bb.src_index, keep, res.new_pos, res.new_ncig, slot, res.new_cig); wr.close()
        stages["reencode_deflate_write_ms"] = (time.perf_counter() - t0) * 1e3
        eng.close(); src.close()
        out["bam_stages_serial_ms"] = {k: round(v, 1) for k, v in stages.items()}
        out["bam_stages_note"] = ("each stage alone on the whole %d-read file with this host's threads; the commands walk the file in pieces of "
                                  "4 MB (compressed) and run inflate of piece k+1, decode + GPU of piece k and re-encode + deflate of piece k-1 "
                                  "side by side, so a command costs about its longest stage plus start-up" % nb)
    except Exception as ex:
        out["bam_stages_error"] = "%s: %s" % (type(ex).__name__, ex)
    log = sys.stderr
    sys.stderr = open(os.devnull, "w")        # the commands log progress lines like the reference does
    try:
        t_aio = t_var = 1e9
        for it in range(2):
            outs = {k: os.path.join(tmp, "%s%d.%s" % (k, it, ext)) for k, ext in (("t", "bam"), ("v", "vcf"), ("c", "fas"))}
            t0 = time.perf_counter()
            amplipy.main(["aio", "-i", inp, "-p", os.path.join(tmp, "p.bed"), "-r", os.path.join(tmp, "ref.fas"),
                          "-ot", outs["t"], "-ov", outs["v"], "-oc", outs["c"]])
            t_aio = min(t_aio, time.perf_counter() - t0)
            t0 = time.perf_counter()
            amplipy.main(["variants", "-i", inp, "-r", os.path.join(tmp, "ref.fas"), "-o", os.path.join(tmp, "vv%d.vcf" % it)])
            t_var = min(t_var, time.perf_counter() - t0)
    finally:
        sys.stderr.close()
        sys.stderr = log
    out["bam_to_bam_reads_per_s"] = round(nb / t_aio, 1)
    out["bam_to_calls_reads_per_s"] = round(nb / t_var, 1)
    out["bam_compression_ratio"] = round(raw_bytes / max(os.path.getsize(inp), 1), 2)
    out["bam_sample"] = ("%d-read BAM of DISTINCT records (%.1f MB, %.1f x smaller than its records: random qualities as in a real run) of the same "
                         "workload; whole `aio` (trimmed BAM + VCF + consensus) and `variants` commands, best of 2, zlib level of the writer as shipped"
                         % (nb, os.path.getsize(inp) / 1e6, raw_bytes / max(os.path.getsize(inp), 1)))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out
