"""Parity of the HIP path (through the C ABI) with the reference-derived golden vectors and
with the CPU oracle.  Integer / index work: the bar is bit-exact."""
import numpy as np
import pytest

from amplipy_amd import synth
from amplipy_amd.batch import ReadBatch
from oracle import oracle
from tests import helpers as H
from tests.gpu_util import GpuRunner, assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[7, 6, 4, 5, 2, 1, 3], ids=["fast_kernel_v7", "fast_kernel_v6", "fast_kernel", "fast_kernel_v5", "tile_kernel", "lane_kernel", "split_pipeline"])
def runner(request):
    r = GpuRunner(variant=request.param)
    yield r
    r.close()


@pytest.fixture(scope="module")
def scheme():
    g = synth.make_genome()
    primers, amps = synth.make_artic_scheme()
    pr = [(s, e) for s, e, _ in primers]
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, 0)
    return g, pr, amps, mn, mx, mpl


@pytest.mark.parametrize("fixture", ["named_cases.json", "random_reads.json.gz"])
def test_golden_per_read(runner, fixture):
    def fn(b, G, mn, mx, mpl, mq, w, do_trim):
        return runner.process(b, G, mn, mx, mpl, mq, w, do_trim=do_trim)
    fails = []
    for case in H.load_json(fixture)["cases"]:
        fails += H.check_case_with(fn, case, oracle.find_overlapping_primers)
    assert not fails, "\n".join(fails[:40])


def test_golden_pileup_batch(runner):
    g = H.load_json("pileup_5000.json.gz")
    b = ReadBatch.from_segments([H.seg_from_dict(d) for d in g["reads"]])
    mn, mx, mpl = oracle.find_overlapping_primers(g["ref_len"], g["primers"], g["offset"])
    r = runner.process(b, g["ref_len"], mn, mx, mpl, g["params"]["min_quality"], g["params"]["window"])
    assert not r.trim.status.any()
    for i, (pos, cig, flags, reflen) in enumerate(g["trim"]):
        assert (int(r.trim.new_pos[i]), r.trim.cigar_string(i), int(r.trim.ref_len[i])) == (pos, cig, reflen), i
        assert [bool(r.trim.trim_flags[i] & m) for m in (1, 2, 4)] == flags
    assert H.sparse_from_engine(r.counts, b, r.events) == H.sparse_from_golden(g["counts"])


@pytest.mark.parametrize("seed,mq,w,off", [(1, 20, 4, 0), (2, 0, 1, 1), (3, 30, 9, 2), (4, 25, 200, 0), (5, 20, 4, 5)])
def test_random_reads_vs_oracle(runner, seed, mq, w, off):
    g = synth.make_genome()
    primers, _ = synth.make_artic_scheme()
    pr = [(s, e) for s, e, _ in primers]
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, off)
    rng = np.random.default_rng(seed)
    segs = synth.random_segments(rng, 6000, g.size, pr)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, g.size, mn, mx, mpl, mq, w)
    d = runner.process(b, g.size, mn, mx, mpl, mq, w)
    assert_same(a, d, b, check_counts=False)
    ok = np.nonzero(a.trim.status == 0)[0]
    good = ReadBatch.from_segments([segs[i] for i in ok])
    for do_trim in (True, False):
        a = oracle.process(good, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
        bad = a.trim.status != 0
        if bad.any():  # reads that only fail when counted untrimmed
            good2 = ReadBatch.from_segments([segs[ok[i]] for i in np.nonzero(~bad)[0]])
            a = oracle.process(good2, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
            d = runner.process(good2, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
            assert_same(a, d, good2)
        else:
            d = runner.process(good, g.size, mn, mx, mpl, mq, w, do_trim=do_trim)
            assert_same(a, d, good)


def test_config2_1k_depth(runner, scheme):
    """BASELINE config 2: 29.9 kb genome, 150 bp reads at 1k x (199,353 reads)."""
    g, pr, amps, mn, mx, mpl = scheme
    b = synth.make_amplicon_batch(g, amps, synth.reads_for_depth(1000), seed=1)
    a = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
    d = runner.process(b, g.size, mn, mx, mpl, 20, 4)
    assert_same(a, d, b)


def test_config5_mixed_pool(runner, scheme):
    """BASELINE config 5 shapes: 75-300 bp, long soft clips, indel-heavy CIGARs."""
    g, pr, amps, mn, mx, mpl = scheme
    segs = synth.make_mixed_segments(g, amps, 20000, seed=3)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
    d = runner.process(b, g.size, mn, mx, mpl, 20, 4)
    assert_same(a, d, b)


def test_empty_and_tiny_batches(runner, scheme):
    g, pr, amps, mn, mx, mpl = scheme
    empty = ReadBatch.from_segments([])
    d = runner.process(empty, g.size, mn, mx, mpl)
    assert d.counts.sum() == 0 and d.events.size == 0
    b = synth.make_amplicon_batch(g, amps, 1, seed=9)
    assert_same(oracle.process(b, g.size, mn, mx, mpl), runner.process(b, g.size, mn, mx, mpl), b)
    b = synth.make_amplicon_batch(g, amps, 65, seed=10)
    assert_same(oracle.process(b, g.size, mn, mx, mpl), runner.process(b, g.size, mn, mx, mpl), b)


def test_sharding_is_exact(runner, scheme):
    """Counts are integer sums over reads: any partition of the batch gives the same table."""
    g, pr, amps, mn, mx, mpl = scheme
    b = synth.make_amplicon_batch(g, amps, 50000, seed=11)
    whole = runner.process(b, g.size, mn, mx, mpl)
    e = runner.engine(g.size)
    e.reset()
    cuts = [0, 7, 12345, 12346, 40000, b.n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e.process(b.slice(lo, hi), read_base=lo)
    assert np.array_equal(e.counts(), whole.counts)
    from tests.gpu_util import EV_ORDER
    assert np.array_equal(np.sort(e.events(), order=EV_ORDER), np.sort(whole.events, order=EV_ORDER))


def test_replication_property_full_depth(runner, scheme):
    """Size-independent check at BASELINE's 100k x size: a batch repeated k times must give
    exactly k times the counts of one copy (which is itself checked against the oracle)."""
    g, pr, amps, mn, mx, mpl = scheme
    base = synth.make_amplicon_batch(g, amps, synth.reads_for_depth(1000), seed=2)
    a = oracle.process(base, g.size, mn, mx, mpl)
    e = runner.engine(g.size)
    e.reset(); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
    k = 100 if runner.variant in (4, 5, 6, 7) else 10
    for rep in range(k):
        e.process(base, read_base=0, want_trim=False)
    assert np.array_equal(e.counts(), a.counts * np.uint32(k))
    assert e.events().size == a.events.size * k


def _golden_calls_check(runner, g, batch, do_trim, mn=None, mx=None, mpl=0):
    from amplipy_amd import calling
    from amplipy_amd.insertions import event_strings
    pr = g["params"]
    r = runner.process(batch, g["ref_len"], mn, mx, mpl, pr["min_quality"], pr.get("window", 4), do_trim=do_trim)
    assert not r.trim.status.any()
    assert H.sparse_from_engine(r.counts, batch, r.events) == H.sparse_from_golden(g["counts"])
    e = runner.engine(g["ref_len"])
    ref_seq = g.get("ref_seq") or synth.genome_string(synth.make_genome())
    e.set_reference(ref_seq)
    pairs = event_strings(batch, r.events)
    provider = lambda positions: calling.tallies_from_events(pairs, positions)
    want = {c["pos"]: c for c in g["calls"]}
    for full in (True, False):
        cp = calling.call_params(pr["min_depth_consensus"], pr["min_freq_consensus"], pr["min_depth_variants"],
                                 pr["min_freq_variants"], True, True, full_ranking=full)
        res = calling.call(e, ref_seq, cp, provider, want_alleles=full)
        got_rec = {v.pos: v.as_dict() for v in res.records}
        cons = res.consensus
        assert res.consensus_string("N") == "".join(c if c is not None else "N" for c in cons)
        for p in range(g["ref_len"]):
            w = want.get(p)
            assert cons[p] == (w.get("consensus") if w else None), (p, full)
            assert got_rec.get(p) == (w.get("variant") if w else None), (p, full)
        if full:
            assert set(res.alleles) == set(want)
            for p, w in want.items():
                total, ranked = res.alleles[p]
                assert total == w["total"]
                assert [[c, float(f).hex(), k] for c, f, k in ranked] == w["alleles"], p
        else:
            assert res.n_relevant <= len(want)


def test_calls_match_golden_trimmed(runner):
    g = H.load_json("pileup_5000.json.gz")
    b = ReadBatch.from_segments([H.seg_from_dict(d) for d in g["reads"]])
    mn, mx, mpl = oracle.find_overlapping_primers(g["ref_len"], g["primers"], g["offset"])
    _golden_calls_check(runner, g, b, True, mn, mx, mpl)


def test_calls_match_golden_untrimmed(runner):
    g = H.load_json("pileup_notrim_1500.json.gz")
    reads = H.load_json("pileup_5000.json.gz")["reads"][:g["n_reads"]]
    b = ReadBatch.from_segments([H.seg_from_dict(d) for d in reads])
    _golden_calls_check(runner, g, b, False)


def test_device_double_division_is_python_division(runner):
    """The frequency thresholds compare IEEE doubles count/total; the device must agree with
    Python on ties exactly at the threshold."""
    from amplipy_amd import calling
    G = 4096
    e = runner.engine(G)
    e.reset()
    rng = np.random.default_rng(5)
    counts = np.zeros((G, 6), np.uint32)
    counts[:, 0] = rng.integers(1, 5000, size=G)      # A
    counts[:, 1] = rng.integers(0, 5000, size=G)      # C
    counts[:, 3] = rng.integers(0, 50, size=G)        # T
    e.add_counts(counts)
    ref = "A" * G
    e.set_reference(ref)
    tot = counts.sum(axis=1).astype(np.int64)
    for thr_pos in (7, 100, 2222):
        f = float(counts[thr_pos, 1]) / float(tot[thr_pos]) if counts[thr_pos, 1] else 0.25
        cp = calling.call_params(1, f, 1, f, True, True)
        res = calling.call(e, ref, cp, None)
        rec = {v.pos: v for v in res.records}
        cons = res.consensus
        for p in range(G):
            alts = [s for c, s in sorted(((int(counts[p, k]), "ACGTN-"[k]) for k in range(6) if counts[p, k]), reverse=True)
                    if s != "A" and c / int(tot[p]) >= f]
            assert (rec[p].alts if p in rec else []) == alts, p
            top = max(((int(counts[p, k]), "ACGTN-"[k]) for k in range(6) if counts[p, k]))
            assert cons[p] == (top[1] if top[0] / int(tot[p]) >= f else None), p


@pytest.mark.parametrize("ext", ["sam", "bam"])
def test_end_to_end_aio_cli(tmp_path, ext, runner, monkeypatch):
    """run_amplipy 'aio' on files: trimmed reads, VCF and consensus agree with the golden pileup."""
    if runner.variant != 4:
        pytest.skip("the CLI always uses the default kernel")
    from amplipy_amd import amplipy, bamio
    g = H.load_json("pileup_5000.json.gz")
    ref = tmp_path / "ref.fas"; ref.write_text(">SYN_REF test\n" + g["ref_seq"] + "\n")
    bed = tmp_path / "p.bed"; bed.write_text("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e, i) for i, (s, e) in enumerate(g["primers"])))
    hdr = bamio.Header("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:SYN_REF\tLN:%d\n@PG\tID:sim\tPN:sim\n" % g["ref_len"], [("SYN_REF", g["ref_len"])])
    inp = str(tmp_path / ("in." + ext))
    w = bamio.AlignmentWriter(inp, "wb" if ext == "bam" else "w", hdr)
    for i, d in enumerate(g["reads"]):
        qual = bytes(ord(c) - 33 for c in d["qual"])
        from amplipy_amd.segment import parse_cigar
        w.write(bamio.Rec("r%d" % i, d["flag"], 0, d["pos"], 60, parse_cigar(d["cigar"]), 0, d["pos"], d["tlen"], d["seq"], qual, aux_sam=["NM:i:1"]))
    w.close()
    out_t = str(tmp_path / ("t." + ext)); out_v = str(tmp_path / "v.vcf"); out_c = str(tmp_path / "c.fas")
    p = g["params"]
    amplipy.main(["aio", "-i", inp, "-p", str(bed), "-r", str(ref), "-ot", out_t, "-ov", out_v, "-oc", out_c,
                  "-mq", str(p["min_quality"]), "-s", str(p["window"]), "-ml", str(p["min_length"]),
                  "-mdc", str(p["min_depth_consensus"]), "-mfc", str(p["min_freq_consensus"]),
                  "-mdv", str(p["min_depth_variants"]), "-mfv", str(p["min_freq_variants"])])
    # trimmed reads: exactly the reads passing AmpliPy.py:910, with the golden POS / CIGAR
    got = [(r.qname, r.pos, r.cigar) for r in bamio.AlignmentReader(out_t, "rb" if ext == "bam" else "r")]
    want = [("r%d" % i, pos, parse_cigar(cig)) for i, (pos, cig, fl, rl) in enumerate(g["trim"])
            if rl >= p["min_length"] and (fl[0] or fl[1])]
    assert got == want
    hdr_out = bamio.AlignmentReader(out_t, "rb" if ext == "bam" else "r").header
    assert hdr_out.to_dict_pg()[-1]["ID"] == "AmpliPy" and hdr_out.to_dict_pg()[-1]["PP"] == "sim"
    # consensus and VCF
    calls = {c["pos"]: c for c in g["calls"]}
    cons = "".join(calls[q]["consensus"] if q in calls and "consensus" in calls[q] else "N" for q in range(g["ref_len"]))
    assert open(out_c).read() == ">sample\n%s\n" % cons
    recs = [l.rstrip("\n").split("\t") for l in open(out_v) if not l.startswith("#")]
    wantv = [c for c in g["calls"] if "variant" in c]
    assert len(recs) == len(wantv)
    for f, c in zip(recs, wantv):
        v = c["variant"]
        info = dict(kv.split("=") for kv in f[7].split(";"))
        assert (f[0], int(f[1]), f[3], f[4], f[6]) == ("SYN_REF", c["pos"] + 1, v["ref"], ",".join(v["alts"]), "PASS")
        assert (int(info["DP"]), int(info["REF_DP"]), info["ALT_DP"], info["ALT_FREQ"]) == (v["DP"], v["REF_DP"], v["ALT_DP"], v["ALT_FREQ"])
        assert info["REF_FREQ"] == "%g" % float.fromhex(v["REF_FREQ"]) and f[9] == "/".join(str(x) for x in v["GT"])
    if ext == "bam":
        # the run above went through libampbam (BAM in, BAM out); the Python codec must give the same stream
        import gzip
        # ... and the same file walked in many small pieces (pieces opened ahead on a thread, rows written behind, buffers
        # handed from piece to piece through the codec's pool) must give the same three outputs
        out_t3 = str(tmp_path / "t3.bam"); out_v3 = str(tmp_path / "v3.vcf"); out_c3 = str(tmp_path / "c3.fas")
        monkeypatch.setenv("AMPLIPY_PART_BYTES", str(48 << 10))
        assert amplipy.native_parts(inp)[0] >= 4
        amplipy.main(["aio", "-i", inp, "-p", str(bed), "-r", str(ref), "-ot", out_t3, "-ov", out_v3, "-oc", out_c3,
                      "-mq", str(p["min_quality"]), "-s", str(p["window"]), "-ml", str(p["min_length"]),
                      "-mdc", str(p["min_depth_consensus"]), "-mfc", str(p["min_freq_consensus"]),
                      "-mdv", str(p["min_depth_variants"]), "-mfv", str(p["min_freq_variants"])])
        monkeypatch.delenv("AMPLIPY_PART_BYTES")
        assert gzip.decompress(open(out_t, "rb").read()) == gzip.decompress(open(out_t3, "rb").read())
        strip = lambda t: [l for l in t.splitlines() if not l.startswith("##source=")]
        assert strip(open(out_v).read()) == strip(open(out_v3).read()) and open(out_c).read() == open(out_c3).read()
        out_t2 = str(tmp_path / "t2.bam"); out_v2 = str(tmp_path / "v2.vcf"); out_c2 = str(tmp_path / "c2.fas")
        monkeypatch.setenv("AMPLIPY_PYTHON_BAM", "1")
        monkeypatch.setattr("sys.argv", list(__import__("sys").argv))
        amplipy.main(["aio", "-i", inp, "-p", str(bed), "-r", str(ref), "-ot", out_t2, "-ov", out_v2, "-oc", out_c2,
                      "-mq", str(p["min_quality"]), "-s", str(p["window"]), "-ml", str(p["min_length"]),
                      "-mdc", str(p["min_depth_consensus"]), "-mfc", str(p["min_freq_consensus"]),
                      "-mdv", str(p["min_depth_variants"]), "-mfv", str(p["min_freq_variants"])])
        assert gzip.decompress(open(out_t, "rb").read()) == gzip.decompress(open(out_t2, "rb").read())
        assert open(out_v).read() == open(out_v2).read() and open(out_c).read() == open(out_c2).read()


# ---- layouts the kernels do not favour: results must not depend on them ------------------------
def _long_read_segments(rng, n, ref_len, max_len, max_ops):
    """Regular CIGARs with many ops (M/=/X alternating with I/D/N), optional clips, lengths up to max_len."""
    from amplipy_amd.segment import Segment
    segs = []
    for _ in range(n):
        n_body = int(rng.integers(1, max_ops // 2 + 1)) * 2 - 1
        L_target = int(rng.integers(50, max_len + 1))
        ops, q = [], 0
        if rng.random() < 0.4:
            k = int(rng.integers(1, 30)); ops.append((4, k)); q += k
        per = max(1, (L_target - q) // ((n_body + 1) // 2))
        for b in range(n_body):
            if b % 2 == 0:
                k = int(rng.integers(1, per + 1)); ops.append((int(rng.choice([0, 0, 7, 8])), k)); q += k
            else:
                op = int(rng.choice([1, 2, 3])); k = int(rng.integers(1, 9)); ops.append((op, k)); q += k if op == 1 else 0
        if rng.random() < 0.4:
            k = int(rng.integers(1, 30)); ops.append((4, k)); q += k
        span = sum(k for o, k in ops if o in (0, 2, 3, 7, 8))
        pos = int(rng.integers(0, max(1, ref_len - span - 1)))
        seq = "".join(rng.choice(list("ACGTN"), q, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        qual = rng.choice([37, 25, 11, 2], q, p=[0.8, 0.12, 0.06, 0.02]).astype(np.uint8)
        segs.append(Segment(flag=int(rng.choice([0, 16, 99, 147])), reference_start=pos, cigar=ops,
                            template_length=int(rng.integers(-500, 500)), query_sequence=seq, query_qualities=qual.tolist()))
    return segs


def test_unsorted_input_gives_the_same_table(runner, scheme):
    """The kernels exploit coordinate order (LDS windows) but must not rely on it."""
    g, pr, amps, mn, mx, mpl = scheme
    segs = synth.make_mixed_segments(g, amps, 6000, seed=21)
    rng = np.random.default_rng(21)
    perm = rng.permutation(len(segs))
    sorted_b = ReadBatch.from_segments(segs)
    shuffled = ReadBatch.from_segments([segs[i] for i in perm])
    a = oracle.process(shuffled, g.size, mn, mx, mpl, 20, 4)
    d = runner.process(shuffled, g.size, mn, mx, mpl, 20, 4)
    assert_same(a, d, shuffled)
    assert np.array_equal(d.counts, runner.process(sorted_b, g.size, mn, mx, mpl, 20, 4).counts)


def test_sparse_coverage_of_a_large_reference(runner):
    """2 Mb reference at < 1 x: every tile moves the LDS window; most adds fall outside it."""
    G = 2_000_000
    rng = np.random.default_rng(22)
    primers = sorted((int(s), int(s) + 25) for s in rng.integers(0, G - 30, 400))
    mn, mx, mpl = oracle.find_overlapping_primers(G, primers, 0)
    segs = synth.random_segments(rng, 8000, G, primers, domain_errors=False)
    segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, mpl, 20, 4)
    ok = np.nonzero(a.trim.status == 0)[0]
    good = ReadBatch.from_segments([segs[i] for i in ok])
    assert good.n > 7000
    assert_same(oracle.process(good, G, mn, mx, mpl, 20, 4), runner.process(good, G, mn, mx, mpl, 20, 4), good)


@pytest.mark.parametrize("max_len,max_ops,n", [(600, 16, 3000), (3000, 24, 1500), (9000, 120, 300)])
def test_long_reads_with_many_ops(runner, max_len, max_ops, n):
    """More CIGAR ops than the tile kernel's column (5), than the second pass's LDS columns (17), and reads
    of thousands of bases: all take the deferred paths and must still be exact."""
    G = 60_000
    rng = np.random.default_rng(max_ops)
    primers = sorted((int(s), int(s) + int(rng.integers(18, 31))) for s in rng.integers(0, G - 40, 120))
    mn, mx, mpl = oracle.find_overlapping_primers(G, primers, 0)
    segs = _long_read_segments(rng, n, G, max_len, max_ops)
    segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, mpl, 20, 4)
    ok = np.nonzero(a.trim.status == 0)[0]
    assert len(ok) > 0.9 * n
    good = ReadBatch.from_segments([segs[i] for i in ok])
    a = oracle.process(good, G, mn, mx, mpl, 20, 4)
    assert_same(a, runner.process(good, G, mn, mx, mpl, 20, 4), good)
    assert int(np.diff(good.cig_off).max()) > min(max_ops - 4, 17)
    # the statuses of the failing reads, too
    d_all = runner.process(b, G, mn, mx, mpl, 20, 4)
    assert np.array_equal(d_all.trim.status, a_all_status(b, G, mn, mx, mpl))


def a_all_status(b, G, mn, mx, mpl):
    return oracle.process(b, G, mn, mx, mpl, 20, 4).trim.status


def test_one_very_long_read(runner):
    """l_seq >= 65536 does not fit the tile kernel's 16-bit query offsets: exact serial path."""
    from amplipy_amd.segment import Segment
    G = 200_000
    rng = np.random.default_rng(5)
    L = 70_000
    ops = [(4, 10), (0, 30_000), (2, 5), (0, 20_000), (1, 4), (0, L - 50_014), ]
    seq = "".join(rng.choice(list("ACGT"), L)); qual = rng.choice([37, 25, 11], L, p=[0.9, 0.07, 0.03]).tolist()
    s_long = Segment(flag=0, reference_start=1000, cigar=ops, template_length=0, query_sequence=seq, query_qualities=qual)
    mn = np.full(G, -1, np.int32); mx = mn.copy(); mn[1000:1030] = 1000; mx[1000:1030] = 1030
    short = _long_read_segments(rng, 200, G, 300, 6)
    segs = sorted(short + [s_long], key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, 30, 20, 4)
    ok = np.nonzero(a.trim.status == 0)[0]
    good = ReadBatch.from_segments([segs[i] for i in ok])
    assert int(good.lseq.max()) == L
    assert_same(oracle.process(good, G, mn, mx, 30, 20, 4), runner.process(good, G, mn, mx, 30, 20, 4), good)


def test_function_level_seams_replay_the_named_cases():
    """amplipy_amd.compat: trim_read / update_base_counts / alleles_from_counts / find_overlapping_primers with
    the reference's own signatures, replayed on golden named cases (one GPU call per read)."""
    from collections import Counter
    from amplipy_amd import compat
    cases = H.load_json("named_cases.json")["cases"]
    n_checked = 0
    for case in cases:
        G = case["ref_len"]
        mn, mx = compat.find_overlapping_primers(G, [tuple(p) for p in case["primers"]], case["offset"])
        assert len(mn) == G and all(v is None or isinstance(v, int) for v in mn[:50])
        for rd, exp in list(zip(case["reads"], case["expected"]))[:8]:
            t = exp["trim"]
            s = H.seg_from_dict(rd)
            if "error" in t:
                with pytest.raises(Exception) as ei:
                    compat.trim_read(s, mn, mx, case["max_primer_len"], case["min_quality"], case["window"])
                assert type(ei.value).__name__ == t["error"]
                continue
            flags = compat.trim_read(s, mn, mx, case["max_primer_len"], case["min_quality"], case["window"])
            assert (s.reference_start, s.cigarstring, list(flags)) == (t["pos"], t["cigar"], t["flags"])
            ct = exp["count_trimmed"]
            table = [dict() for _ in range(G)]
            if "error" in ct:
                with pytest.raises(Exception) as ei:
                    compat.update_base_counts(table, s, case["min_quality"])
                assert type(ei.value).__name__ == ct["error"]
                continue
            compat.update_base_counts(table, s, case["min_quality"])
            got = Counter({(p, k): n for p, d in enumerate(table) for k, n in d.items()})
            assert got == H.sparse_from_golden(ct["counts"])
            n_checked += 1
    assert n_checked >= 30
    tot, alleles = compat.alleles_from_counts({"A": 3, "C": 0, "AT": 3, "-": 1, "T": 3})
    assert tot == 10 and [a[2] for a in alleles] == ["T", "AT", "A", "-"] and alleles[0][1] == 0.3


@pytest.mark.parametrize("seed,mq,w,off,lmin,lmax", [(61, 20, 4, 0, 160, 304), (62, 20, 4, 1, 100, 200), (63, 10, 4, 0, 240, 304), (64, 30, 3, 2, 60, 304), (65, 20, 4, 0, 250, 250)])
def test_reads_of_up_to_304_bases_with_one_indel(runner, seed, mq, w, off, lmin, lmax):
    """The second-generation fast kernel (amp_fast5.hpp, branch-free closed forms of amp_bf.hpp) takes reads of up to 304
    bases and runs in three builds chosen by the batch's mean padded read length (runs of 9.7 / 13.3 / 19.5 KB per tile,
    packed windows of 256 / 512 positions): reads of the given length range with one match op or two around one insertion /
    deletion, soft clips, dense primer tables, low-quality ends, N calls; reads longer than the fast path takes mixed in.
    Trim results, count table and insertion events against the oracle (every kernel variant must agree)."""
    from amplipy_amd.segment import Segment
    rng = np.random.default_rng(seed)
    G = 9000
    primers = sorted((int(a), int(a) + int(rng.integers(18, 32))) for a in rng.integers(0, G - 40, 80))
    mn, mx, mpl = oracle.find_overlapping_primers(G, primers, off)
    segs = []
    starts = np.sort(np.concatenate([rng.integers(0, G - 400, 40), [0, 1, G - 330]]))
    for s0 in starts:
        for _ in range(int(rng.integers(40, 160))):
            L = int(rng.integers(lmin, lmax + 1)) if rng.random() < 0.97 else int(rng.integers(305, 340))
            kind = int(rng.integers(0, 3)) if rng.random() < 0.4 else 0
            sa = int(rng.integers(1, 30)) if rng.random() < 0.2 else 0
            sc = int(rng.integers(1, 30)) if rng.random() < 0.2 else 0
            k = int(rng.integers(1, 9)) if kind else 0
            body = L - sa - sc - (k if kind == 1 else 0)
            if body < 4:
                continue
            m1 = int(rng.integers(1, body - 1)) if kind else body
            m2 = body - m1 if kind else 0
            cig = ([(4, sa)] if sa else []) + [(0, m1)] + ([(kind, k), (0, m2)] if kind else []) + ([(4, sc)] if sc else [])
            span = m1 + m2 + (k if kind == 2 else 0)
            pos = int(min(max(s0 + rng.integers(-3, 4), 0), G - span - 1))
            q = rng.choice([37, 25, 11, 2], L, p=[0.7, 0.15, 0.1, 0.05]).astype(np.int64)
            t = int(rng.integers(0, 40))
            if rng.random() < 0.3:
                q[:t] = 2
            elif rng.random() < 0.4:
                q[L - t:] = 2
            seq = "".join(rng.choice(list("ACGTN"), L, p=[0.2475, 0.2475, 0.2475, 0.2475, 0.01]))
            flag = int(rng.choice([0, 16, 99, 147, 83, 163]))
            segs.append(Segment(flag=flag, reference_start=pos, cigar=cig, template_length=int(rng.choice([0, 400, -400, 90])),
                                query_sequence=seq, query_qualities=q.tolist()))
    segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, mpl, mq, w)
    d = runner.process(b, G, mn, mx, mpl, mq, w)
    assert_same(a, d, b, check_counts=False)
    ok = np.nonzero(a.trim.status == 0)[0]
    assert ok.size > 0.9 * b.n
    good = ReadBatch.from_segments([segs[i] for i in ok])
    for do_trim in (True, False):
        a = oracle.process(good, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        keep = np.nonzero(a.trim.status == 0)[0]
        good2 = good if keep.size == good.n else ReadBatch.from_segments([segs[ok[i]] for i in keep])
        a = oracle.process(good2, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        d = runner.process(good2, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        assert_same(a, d, good2)
        assert a.events.size > 0
    if runner.variant == 5 and w == 4:
        # the fast path really took them: what is left for the general pass is the reads of more than 304 bases and little else
        left = int(runner.engine(G).debug_counters()[7])
        assert left < 0.1 * good2.n, (left, good2.n)


@pytest.mark.parametrize("seed,mq,w,off", [(41, 20, 4, 0), (42, 10, 1, 2), (43, 30, 8, 1), (44, 0, 3, 0), (45, 25, 6, 3), (46, 20, 2, 0)])
def test_short_reads_with_one_indel(runner, seed, mq, w, off):
    """The shapes the fast kernel does in closed form (amp_fast.hpp / Cig2 in amp_read.hpp): reads of up to 152 bases with one
    match op or two around ONE insertion / deletion, with or without soft clips at the ends, dense primer tables so that the clips of A:450-558 and A:589-686 often
    stop inside or right in front of the indel, low-quality runs inside insertions (several events per insertion), reads
    at reference position 0 (A:735-736), at the end of the reference, with N calls of good quality, and piles further
    apart than the kernel's packed window.  Trim results, count table and insertion events against the oracle."""
    from amplipy_amd.segment import Segment
    rng = np.random.default_rng(seed)
    G = 6000
    primers = sorted((int(a), int(a) + int(rng.integers(18, 32))) for a in rng.integers(0, G - 40, 60))
    mn, mx, mpl = oracle.find_overlapping_primers(G, primers, off)
    segs = []
    starts = np.sort(np.concatenate([rng.integers(0, G - 200, 40), [0, 0, 1, G - 160]]))
    for s0 in starts:
        for _ in range(int(rng.integers(20, 140))):                      # a pile of reads per start, jittered
            kind = int(rng.integers(0, 3))
            m1 = int(rng.integers(1, 100)); k = int(rng.integers(1, 9)) if kind else 0; m2 = int(rng.integers(1, 60)) if kind else 0
            sa = int(rng.integers(1, 25)) if rng.random() < 0.3 else 0      # soft clips at the ends, as aligners leave them
            sc = int(rng.integers(1, 25)) if rng.random() < 0.3 else 0
            L = sa + m1 + (k if kind == 1 else 0) + m2 + sc
            if L > 152:
                continue
            op = 7 if rng.random() < 0.1 else 0
            cig = ([(4, sa)] if sa else []) + [(op, m1)] + ([(kind, k), (op, m2)] if kind else []) + ([(4, sc)] if sc else [])
            pos = int(min(max(s0 + rng.integers(-3, 4), 0), G - (m1 + m2 + (k if kind == 2 else 0)) - 1))
            q = rng.choice([37, 25, 11, 2], L, p=[0.7, 0.15, 0.1, 0.05]).astype(np.int64)
            t = int(rng.integers(0, 25))
            if rng.random() < 0.3:
                q[:t] = 2
            elif rng.random() < 0.4:
                q[L - t:] = 2
            seq = "".join(rng.choice(list("ACGTN"), L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
            flag = int(rng.choice([0, 16, 99, 147, 83, 163]))
            segs.append(Segment(flag=flag, reference_start=pos, cigar=cig, template_length=int(rng.choice([0, 300, -300, 90])),
                                query_sequence=seq, query_qualities=q.tolist()))
    segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, mpl, mq, w)
    d = runner.process(b, G, mn, mx, mpl, mq, w)
    assert_same(a, d, b, check_counts=False)
    ok = np.nonzero(a.trim.status == 0)[0]
    assert ok.size > 0.9 * b.n
    good = ReadBatch.from_segments([segs[i] for i in ok])
    for do_trim in (True, False):
        a = oracle.process(good, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        keep = np.nonzero(a.trim.status == 0)[0]
        good2 = good if keep.size == good.n else ReadBatch.from_segments([segs[ok[i]] for i in keep])
        a = oracle.process(good2, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        d = runner.process(good2, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        assert_same(a, d, good2)
        assert a.events.size > 50


def test_coordinate_helpers_on_the_device():
    """Rows a2-a4 of SURVEY.md section 8 on their own: the 1,500 reference-derived vectors of tests/golden/helpers.json
    (get_pos_on_query A:389-412, get_pos_on_ref A:363-386, fix_cigar A:415-423) replayed through the DEVICE functions
    the kernels use (amp_coordinate_helpers), in one batch and through the reference's own signatures (compat)."""
    from amplipy_amd import compat, lib
    cases = H.load_json("helpers.json")["cases"]
    eng = lib.Engine(1000)
    cigs = [[tuple(x) for x in c["cigar"]] for c in cases]
    start = [c["start"] for c in cases]
    oq, orf, fixed, st = eng.coordinate_helpers(cigs, start, [c["start"] + c["x"] for c in cases], [c["x"] for c in cases])
    assert not st.any()
    assert oq.tolist() == [c["pos_on_query"] for c in cases]
    assert orf.tolist() == [c["pos_on_ref"] for c in cases]
    assert fixed == [[tuple(x) for x in c["fix_cigar"]] for c in cases]
    eng.close()
    for c in cases[:25]:
        cig = [tuple(x) for x in c["cigar"]]
        assert compat.get_pos_on_query(cig, c["start"] + c["x"], c["start"]) == c["pos_on_query"]
        assert compat.get_pos_on_ref(cig, c["x"], c["start"]) == c["pos_on_ref"]
        assert compat.fix_cigar(cig) == [tuple(x) for x in c["fix_cigar"]]
    # an op code outside the reference's tables raises there (KeyError on CONSUME_*): reported as a status here
    _, _, _, st = lib.Engine(10).coordinate_helpers([[(0, 5), (9, 2)]], [0], [7], [6])
    assert st[0] != 0


@pytest.mark.parametrize("seed", list(range(3000, 3012)))
def test_randomised_parameters_and_read_shapes(runner, seed):
    """A slice of tools/fuzz_gpu.py: random min_quality / window / primer offset, three families of reads
    (adversarial, indel-heavy mixed, many-op long), sorted or not; trims, counts and events equal the oracle."""
    g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
    pr = [(s, e) for s, e, _ in primers]
    rng = np.random.default_rng(seed)
    mq = int(rng.choice([0, 2, 13, 20, 30, 41])); w = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 30])); off = int(rng.integers(0, 8))
    mn, mx, mpl = oracle.find_overlapping_primers(g.size, pr, off)
    kind = seed % 3
    if kind == 0:
        segs = synth.random_segments(rng, 3000, g.size, pr)
    elif kind == 1:
        segs = synth.make_mixed_segments(g, amps, 3000, seed=seed)
    else:
        segs = _long_read_segments(rng, 800, int(g.size), int(rng.choice([300, 1200, 5000])), int(rng.choice([8, 14, 18, 22, 40, 90])))
    if rng.random() < 0.5:
        segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, g.size, mn, mx, mpl, mq, w)
    assert_same(a, runner.process(b, g.size, mn, mx, mpl, mq, w), b, check_counts=False)
    ok = np.nonzero(a.trim.status == 0)[0]
    good = ReadBatch.from_segments([segs[i] for i in ok])
    assert_same(oracle.process(good, g.size, mn, mx, mpl, mq, w), runner.process(good, g.size, mn, mx, mpl, mq, w), good)


def _wild_long_segments(rng, n, ref_len):
    """CIGARs of 18-160 ops drawn op by op: mostly regular bodies, but also clips inside the alignment, hard clips, pads,
    runs of the same op, insertions at either end or in front of a deletion / the end clip, reads that start on position 0."""
    from amplipy_amd.segment import Segment
    segs = []
    for _ in range(n):
        n_ops = int(rng.integers(18, 161))
        wild = rng.random() < 0.25
        ops, q = [], 0
        if rng.random() < 0.15 and wild: ops.append((5, int(rng.integers(1, 9))))
        if rng.random() < 0.5:
            k = int(rng.integers(1, 25)); ops.append((4, k)); q += k
        prev = -1
        while len(ops) < n_ops:
            if wild:
                op = int(rng.choice([0, 0, 1, 2, 3, 4, 6, 7, 8, 1, 2]))
            else:
                op = int(rng.choice([0, 7, 8])) if prev not in (0, 7, 8) and rng.random() < 0.85 else int(rng.choice([1, 1, 2, 2, 3]))
                if op == prev: continue
            k = int(rng.integers(1, 12)) if op not in (0, 7, 8) else int(rng.integers(1, 40))
            ops.append((op, k)); prev = op
            if op in (0, 1, 4, 7, 8): q += k
        if rng.random() < 0.5:
            k = int(rng.integers(1, 25)); ops.append((4, k)); q += k
        if rng.random() < 0.1 and wild: ops.append((5, 3))
        span = sum(k for o, k in ops if o in (0, 2, 3, 7, 8))
        pos = 0 if rng.random() < 0.03 else int(rng.integers(0, max(1, ref_len - span - 1)))
        seq = "".join(rng.choice(list("ACGTN"), q, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        qual = rng.choice([37, 25, 11, 2], q, p=[0.93, 0.05, 0.015, 0.005] if rng.random() < 0.6 else [0.6, 0.2, 0.12, 0.08]).astype(np.uint8)
        segs.append(Segment(flag=int(rng.choice([0, 16, 99, 147])), reference_start=pos, cigar=ops,
                            template_length=int(rng.integers(-700, 700)), query_sequence=seq, query_qualities=qual.tolist()))
    return segs


@pytest.mark.parametrize("seed,mq,w", [(1, 20, 4), (2, 13, 1), (3, 30, 8), (4, 20, 9), (5, 2, 3), (6, 25, 5)])
def test_many_op_reads_of_every_shape(runner, seed, mq, w):
    """The wave-per-read path (amp_wave.hpp) and the cases it hands to the serial code: statuses of all reads, then trims,
    counts and events of the reads the reference accepts."""
    G = 40_000
    rng = np.random.default_rng(seed)
    primers = sorted((int(s), int(s) + int(rng.integers(18, 31))) for s in rng.integers(0, G - 40, 160))
    mn, mx, mpl = oracle.find_overlapping_primers(G, primers, int(rng.integers(0, 4)))
    segs = _wild_long_segments(rng, 1500, G)
    segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    a = oracle.process(b, G, mn, mx, mpl, mq, w)
    assert_same(a, runner.process(b, G, mn, mx, mpl, mq, w), b, check_counts=False)
    ok = np.nonzero(a.trim.status == 0)[0]
    assert len(ok) > 0.5 * len(segs)
    good = ReadBatch.from_segments([segs[i] for i in ok])
    assert_same(oracle.process(good, G, mn, mx, mpl, mq, w), runner.process(good, G, mn, mx, mpl, mq, w), good)
    if seed <= 2:
        # the untrimmed pile-up (A:915 without A:907) and the trims alone
        a = oracle.process(b, G, mn, mx, mpl, mq, w, do_trim=False)
        ok = np.nonzero(a.trim.status == 0)[0]
        good = ReadBatch.from_segments([segs[i] for i in ok])
        assert_same(oracle.process(good, G, mn, mx, mpl, mq, w, do_trim=False), runner.process(good, G, mn, mx, mpl, mq, w, do_trim=False), good)
        assert_same(oracle.process(b, G, mn, mx, mpl, mq, w, do_count=False), runner.process(b, G, mn, mx, mpl, mq, w, do_count=False), b, check_counts=False)


def _awkward_regular_segments(rng, n, ref_len):
    """Regular CIGARs (H* S* body S* H*) whose body is ANY sequence of M/=/X/I/D/N ops of 2 to 6 entries: insertions and
    deletions side by side, first or last in the body, zero-length ops in between; reads at reference position 0, near and
    over the reference's end; a fifth of the qualities below any threshold used."""
    from amplipy_amd.segment import Segment
    segs = []
    while len(segs) < n:
        ops = []
        if rng.random() < 0.1:
            ops.append((5, int(rng.integers(1, 6))))
        if rng.random() < 0.35:
            ops.append((4, int(rng.integers(1, 25))))
        for _ in range(int(rng.integers(2, 7))):
            op = int(rng.choice([0, 0, 0, 1, 1, 2, 2, 3, 7, 8]))
            ops.append((op, 0 if rng.random() < 0.06 else int(rng.integers(1, 40 if op in (0, 7, 8) else 10))))
        if rng.random() < 0.35:
            ops.append((4, int(rng.integers(1, 25))))
        if rng.random() < 0.1:
            ops.append((5, int(rng.integers(1, 6))))
        q = sum(k for o, k in ops if o in (0, 1, 4, 7, 8)); span = sum(k for o, k in ops if o in (0, 2, 3, 7, 8))
        if q == 0 or q > 150:
            continue
        r = rng.random()
        pos = 0 if r < 0.15 else (ref_len - span + int(rng.integers(-3, 6)) if r < 0.3 else int(rng.integers(0, ref_len - span - 1)))
        pos = max(pos, 0)
        seq = "".join(rng.choice(list("ACGTN"), q, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        qual = rng.choice([37, 25, 11, 2], q, p=[0.6, 0.2, 0.12, 0.08]).astype(np.uint8)
        segs.append(Segment(flag=int(rng.choice([0, 16, 99, 147])), reference_start=pos, cigar=ops,
                            template_length=int(rng.integers(-500, 500)), query_sequence=seq, query_qualities=qual.tolist()))
    return segs


@pytest.mark.parametrize("seed,mq,w", [(11, 20, 4), (12, 13, 3), (13, 30, 8)])
def test_regular_cigars_with_awkward_indels(runner, seed, mq, w):
    """What the op-by-op indel walk of the tile kernel (count_regular_ops, amp_read.hpp) has to get right on the device: the
    run logic of A:730-748 on insertions that are followed by a deletion, by another insertion, by the end clip or by nothing,
    that start the body or sit at reference position 0 (A:735), reads that leave the reference (statuses in pair order).
    Statuses of all reads, then trims, count table and events of the reads the reference accepts, with and without trimming."""
    G = 5000
    rng = np.random.default_rng(seed)
    primers = sorted((int(s), int(s) + int(rng.integers(18, 31))) for s in rng.integers(0, G - 40, 40))
    mn, mx, mpl = oracle.find_overlapping_primers(G, primers, int(rng.integers(0, 3)))
    segs = _awkward_regular_segments(rng, 6000, G)
    segs.sort(key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    for do_trim in (True, False):
        a = oracle.process(b, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        assert_same(a, runner.process(b, G, mn, mx, mpl, mq, w, do_trim=do_trim), b, check_counts=False)
        ok = np.nonzero(a.trim.status == 0)[0]
        assert 0.3 * len(segs) < len(ok) < len(segs)          # both kinds of reads are there
        good = ReadBatch.from_segments([segs[i] for i in ok])
        ag = oracle.process(good, G, mn, mx, mpl, mq, w, do_trim=do_trim)
        assert_same(ag, runner.process(good, G, mn, mx, mpl, mq, w, do_trim=do_trim), good)
        assert ag.events.size > 500


def test_cu_share_does_not_change_results(runner, scheme):
    """amp_set_cu_share sizes the fast kernel's grid for a part of the chip (callers with several batches in flight run their
    passes side by side): trims, table and events must not depend on it -- plain reads, indel reads handed to the general
    pass and a batch spread thinly over the reference alike."""
    if runner.variant not in (4, 5, 6, 7):
        pytest.skip("only the fast kernels have a grid to size")
    g, pr, amps, mn, mx, mpl = scheme
    batches = [synth.make_amplicon_batch(g, amps, 60000, seed=8),
               ReadBatch.from_segments(sorted(synth.make_mixed_segments(g, amps, 9000, seed=9), key=lambda s: s.reference_start))]
    eng = runner.engine(g.size)
    try:
        for b in batches:
            a = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
            for share in (1, 2, 3, 16):
                eng.set_cu_share(share)
                assert_same(a, runner.process(b, g.size, mn, mx, mpl, 20, 4), b)
    finally:
        eng.set_cu_share(1)
    with pytest.raises(Exception):
        eng.set_cu_share(0)


def test_event_text_from_the_staged_batch(scheme):
    """amp_event_strings with reads == NULL (the batch the last amp_process_batch left on the device) gives the allele text
    the host-side gather gives (A:736-738), and fails cleanly before any batch."""
    from amplipy_amd import abi, lib
    from amplipy_amd.insertions import EventStore
    g, pr, amps, mn, mx, mpl = scheme
    segs = sorted(synth.make_mixed_segments(g, amps, 4000, seed=77), key=lambda s: s.reference_start)
    b = ReadBatch.from_segments(segs)
    e = lib.Engine(int(g.size)); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
    with pytest.raises(Exception):
        e.event_text(np.zeros(1, abi.INS_EVENT_DTYPE), 0)
    e.process(b, read_base=1000)
    ev = e.drain_events()
    assert ev.size > 100
    rows = ev.copy(); rows["read"] = ev["read"] - 1000
    length, blob = e.event_text(rows, 0)
    a, d = EventStore(), EventStore()
    a.add(b, ev, 1000); d.add_text(ev["ref_pos"], length, blob)
    assert a.pairs() == d.pairs()
    e.close()


def _runs_as_counter(e, runs, read_base):
    """{(ref_pos, allele text): events} of Engine.aggregate_events records (text of the representative events from the device)."""
    from collections import Counter
    from amplipy_amd import abi
    rows = np.zeros(runs.size, abi.INS_EVENT_DTYPE)
    for f in ("ref_pos", "q_from", "q_to"):
        rows[f] = runs[f]
    rows["read"] = runs["read"] - np.uint32(read_base)
    length, blob = e.event_text(rows, 0) if runs.size else (np.zeros(0, np.int64), np.zeros(0, np.uint8))
    raw = blob.tobytes(); off = np.cumsum(length) - length
    out = Counter()
    for k in range(runs.size):
        out[(int(runs["ref_pos"][k]), raw[int(off[k]):int(off[k]) + int(length[k])].decode("ascii"))] += int(runs["count"][k])
    return out


def test_insertion_events_aggregated_on_the_device(scheme):
    """SURVEY 8f row n4 (the dict tally of A:745-748, consumed at A:767-771) on the device: amp_aggregate_ins_events sorts
    the events by (position, allele) and run-length encodes them.  (1) the insertion keys of the reference-derived golden
    pileup; (2) BASELINE config-5 shapes and (3) a batch of many-op reads against the oracle's event list; the plain
    event list still holds the same events afterwards, drain empties it, and calling from the runs gives the golden calls."""
    from collections import Counter
    from amplipy_amd import calling, lib
    from amplipy_amd.insertions import event_strings
    # (1) golden pileup
    gd = H.load_json("pileup_5000.json.gz")
    b = ReadBatch.from_segments([H.seg_from_dict(d) for d in gd["reads"]])
    mn, mx, mpl = oracle.find_overlapping_primers(gd["ref_len"], gd["primers"], gd["offset"])
    e = lib.Engine(gd["ref_len"]); e.set_primers(mn, mx, mpl); e.set_params(gd["params"]["min_quality"], gd["params"]["window"], True, True)
    e.process(b, read_base=5000)
    runs = e.aggregate_events(read_base=5000)
    got = _runs_as_counter(e, runs, 5000)
    want = Counter({(int(p), k): int(n) for p, k, n in gd["counts"] if k not in ("A", "C", "G", "T", "N", "-")})
    assert len(want) > 20 and got == want
    assert runs.size == len(want)                                       # one record per allele
    assert np.all(np.diff(runs["ref_pos"].astype(np.int64)) >= 0)       # sorted by position
    ev = e.events()
    assert Counter(event_strings(b, ev, 5000)) == want                  # the list is untouched ...
    assert e.aggregate_events(read_base=5000, drain=True).size == runs.size and e.events().size == 0 and e.aggregate_events(read_base=5000).size == 0   # ... until drained
    # calls from the device-side runs equal the golden calls (full rankings)
    e.reset(); e.process(b)
    triples = [(p, s_, c) for (p, s_), c in _runs_as_counter(e, e.aggregate_events(), 0).items()]
    e.set_reference(gd["ref_seq"])
    pr = gd["params"]
    cp = calling.call_params(pr["min_depth_consensus"], pr["min_freq_consensus"], pr["min_depth_variants"], pr["min_freq_variants"], True, True, full_ranking=True)
    res = calling.call(e, gd["ref_seq"], cp, lambda positions: calling.tallies_from_runs(triples, positions), want_alleles=True)
    for c in gd["calls"]:
        total, ranked = res.alleles[c["pos"]]
        assert total == c["total"] and [[n_, float(f).hex(), k] for n_, f, k in ranked] == c["alleles"], c["pos"]
    e.close()
    # (2) config-5 shapes, (3) many-op reads: against the oracle's events
    g, pr_, amps, mn, mx, mpl = scheme
    G2 = 60_000
    rng = np.random.default_rng(9)
    primers2 = sorted((int(s_), int(s_) + int(rng.integers(18, 31))) for s_ in rng.integers(0, G2 - 40, 120))
    cases = [(synth.make_mixed_segments(g, amps, 20000, seed=3), int(g.size), (mn, mx, mpl)),
             (_long_read_segments(rng, 1500, G2, 900, 40), G2, oracle.find_overlapping_primers(G2, primers2, 0))]
    for segs, G_, (mn_, mx_, mpl_) in cases:
        segs = sorted(segs, key=lambda s_: s_.reference_start)
        b = ReadBatch.from_segments(segs)
        a = oracle.process(b, G_, mn_, mx_, mpl_, 20, 4)
        keep = np.nonzero(a.trim.status == 0)[0]
        if keep.size != b.n:                                   # (reads the reference raises on record no events)
            b = ReadBatch.from_segments([segs[i] for i in keep])
            a = oracle.process(b, G_, mn_, mx_, mpl_, 20, 4)
        e = lib.Engine(G_); e.set_primers(mn_, mx_, mpl_); e.set_params(20, 4, True, True)
        e.process(b, read_base=7)
        got = _runs_as_counter(e, e.aggregate_events(read_base=7), 7)
        want = Counter(event_strings(b, a.events, 0))
        assert len(want) > 100 and got == want
        e.close()


def test_single_rank_rccl_paths(tmp_path, scheme, monkeypatch):
    """The N > 1 code has to have run before an 8-GPU node shows up: (a) amp_reduce with no communicator is a no-op,
    (b) one-rank RCCL all-reduce of the bound device table through torch.distributed (backend nccl) leaves the
    table unchanged and equal to the oracle's, (c) the command line under AMPLIPY_FORCE_DIST=1 (one-rank
    torchrun rehearsal: range partition, all-reduce, rank 0 writes) gives the same VCF / consensus as the plain run."""
    import os
    import torch
    import torch.distributed as dist
    from amplipy_amd import amplipy, bamio, lib, parallel
    g, pr, amps, mn, mx, mpl = scheme
    b = synth.make_amplicon_batch(g, amps, 30000, seed=21)
    a = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        e = lib.Engine(g.size)
        table = torch.zeros(g.size * 7, dtype=torch.int32, device="cuda:0")
        e.bind_counts(table.data_ptr())
        e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
        e.process(b)
        e.reduce(None, 0)                                   # (a)
        parallel.allreduce_table(dist, table)               # (b)
        torch.cuda.synchronize()
        got = table.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[:g.size * 6].reshape(g.size, 6), a.counts)
        assert int(got[g.size * 6:].sum()) == a.events.size
        e.close()
    finally:
        dist.destroy_process_group()
    # (c)
    ref = tmp_path / "ref.fas"; ref.write_text(">SYN_REF test\n" + synth.genome_string(g) + "\n")
    bed = tmp_path / "p.bed"; bed.write_text("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e2, i) for i, (s, e2) in enumerate(pr)))
    hdr = bamio.Header("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:SYN_REF\tLN:%d\n@PG\tID:sim\tPN:sim\n" % g.size, [("SYN_REF", g.size)])
    from tools.e2e_legs import write_bam
    inp = str(tmp_path / "in.bam")
    write_bam(inp, b.slice(0, 8000), g.size)
    outs = {}
    for tag, force in (("plain", "0"), ("dist", "1")):
        monkeypatch.setenv("AMPLIPY_FORCE_DIST", force)
        monkeypatch.setenv("MASTER_PORT", "29542")
        v, c = str(tmp_path / (tag + ".vcf")), str(tmp_path / (tag + ".fas"))
        amplipy.main(["aio", "-i", inp, "-p", str(bed), "-r", str(ref), "-ot", str(tmp_path / (tag + ".bam")), "-ov", v, "-oc", c])
        outs[tag] = ([l for l in open(v) if not l.startswith("##")], open(c).read())
    assert outs["plain"] == outs["dist"]
    assert len(outs["plain"][0]) > 1
    # (d) the benchmark's multi-rank path with one rank: the all-reduce on a stream of its own under the next step's reads, the
    # strong-scaling slice of ONE job and its bit-identity check (what `--gpus 8 --strong` runs, minus the other seven ranks)
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT="29545", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", AMPLIPY_FORCE_DIST="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--strong", "--depth", "1000", "--steps", "6",
                          "--warmup", "2", "--cpu-passes", "-1", "--no-e2e", "--no-extra"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["config"]["rccl_ranks"] == 1 and line["config"]["collective_on_own_stream"]
    assert line["config"]["strong_check"] and line["config"]["error_reads"] == 0


def test_amp_reduce_with_a_real_rccl_communicator(scheme):
    """amp_reduce (include/amplihip.h) with an ncclComm_t: a one-rank communicator made through RCCL's own C API
    (ncclGetUniqueId / ncclCommInitRank, resolved from the librccl the process loads), then the library's dlsym'd
    ncclReduce (root = 0) and ncclAllReduce (root < 0) with its hard-coded ncclUint32 / ncclSum run on the device table.
    A sum over one rank is the identity: table and insertion tally stay equal to the oracle's."""
    import ctypes as C
    import glob
    import os
    from amplipy_amd import lib
    g, pr, amps, mn, mx, mpl = scheme
    cands = []
    try:
        import torch
        cands += glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
    except ImportError:
        pass
    cands += glob.glob("/opt/rocm/lib/librccl.so*")
    if not cands:
        pytest.skip("no librccl on this box")
    rccl = C.CDLL(cands[0], mode=C.RTLD_GLOBAL)         # global: amp_reduce looks the symbols up in the process first

    class NcclUniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = NcclUniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(NcclUniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    b = synth.make_amplicon_batch(g, amps, 30000, seed=23)
    a = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
    e = lib.Engine(int(g.size))                         # (selects the device before the communicator is made)
    e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
    e.process(b)
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        for root in (0, -1):
            e.reduce(comm.value, root)
            assert np.array_equal(e.counts(), a.counts), "root %d" % root
            assert e.events().size == a.events.size
        # the calls begun before a reduce are cancelled by it (call_pending): begin, reduce, view == fresh view
        from amplipy_amd import calling
        e.set_reference(synth.genome_string(g))
        cp = calling.call_params(10, 0.0, 1, 0.03, True, True)
        e.call_compact_begin(cp)
        e.reduce(comm.value, -1)
        c1 = [x.copy() for x in e.call_compact(cp)]
        c2 = [x.copy() for x in e.call_compact(cp)]
        assert all(np.array_equal(x, y) for x, y in zip(c1, c2))
    finally:
        rccl.ncclCommDestroy(comm)
        e.close()


def test_begun_calls_are_cancelled_by_table_updates(scheme):
    """amp_call_compact_begin, then amp_add_counts: the view must be of the UPDATED table."""
    from amplipy_amd import calling, lib
    g, pr, amps, mn, mx, mpl = scheme
    b = synth.make_amplicon_batch(g, amps, 20000, seed=29)
    e = lib.Engine(int(g.size)); e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
    e.set_reference(synth.genome_string(g))
    cp = calling.call_params(1, 0.0, 1, 0.03, True, True)
    e.process(b)
    once = e.counts()
    e.call_compact_begin(cp)
    e.add_counts(once)                                   # doubles every count: depths double, frequencies stay
    cons, vr, rel = [x.copy() for x in e.call_compact(cp)]
    assert np.array_equal(e.counts(), once * np.uint32(2))
    fresh = lib.Engine(int(g.size)); fresh.set_primers(mn, mx, mpl); fresh.set_params(20, 4, True, True); fresh.set_reference(synth.genome_string(g))
    fresh.process(b); fresh.add_counts(once)
    cons2, vr2, rel2 = [x.copy() for x in fresh.call_compact(cp)]
    assert np.array_equal(cons, cons2) and np.array_equal(vr, vr2) and np.array_equal(rel, rel2)
    assert vr.size and int(vr["total_depth"].max()) % 2 == 0
    e.close(); fresh.close()


def test_fast_path_report(scheme):
    """amp_fast_path_active (include/amplihip.h): windows of 1..8 and min_quality <= 128 run on the fast kernels, anything else on
    the general tile kernel -- with the same results either way (a window of 12 against the oracle)."""
    from amplipy_amd import lib
    g, pr, amps, mn, mx, mpl = scheme
    e = lib.Engine(g.size)
    e.set_primers(mn, mx, mpl)
    e.set_params(20, 4, True, True); assert e.fast_path_active()
    e.set_params(20, 8, True, True); assert e.fast_path_active()
    e.set_params(129, 4, True, True); assert not e.fast_path_active()
    e.set_params(20, 12, True, True); assert not e.fast_path_active()
    b = synth.make_amplicon_batch(g, amps, 20000, seed=33)
    res = e.process(b)
    a = oracle.process(b, g.size, mn, mx, mpl, 20, 12)
    assert np.array_equal(res.new_pos, a.trim.new_pos) and np.array_equal(e.counts(), a.counts)
    assert e.last_kernel_variant() == 2
    e.set_kernel_variant(2); assert not e.fast_path_active()
    e.close()


def test_cli_on_mixed_long_reads_matches_the_general_kernel(tmp_path, scheme, monkeypatch):
    """`aio` on a BAM of mixed 75-300 bp reads with soft clips and indels everywhere (BASELINE config 5 in small), walked in pieces: the
    default choice (the list-driven fast kernel for such batches) writes the files the general tile kernel writes."""
    from amplipy_amd import amplipy, bam_native, synth as sy
    g, pr, amps, mn, mx, mpl = scheme
    b = sy.make_config5_batch(g, amps, rep=2, pool_reads=30000, seed=17)
    ref = tmp_path / "ref.fas"; ref.write_text(">SYN_REF\n" + sy.genome_string(g) + "\n")
    bed = tmp_path / "p.bed"; bed.write_text("".join("SYN_REF\t%d\t%d\tp%d\n" % (s, e, i) for i, (s, e) in enumerate(pr)))
    from tools.e2e_legs import write_bam
    seed = str(tmp_path / "seed.bam"); write_bam(seed, b.slice(0, 8), g.size)
    inp = str(tmp_path / "in.bam")
    sf = bam_native.BamFile(seed)
    w = bam_native.BamWriter(inp, sf.header_text, sf, level=1); w.write_batch(b); w.close(); sf.close()
    monkeypatch.setenv("AMPLIPY_PART_BYTES", str(1 << 20))
    outs = {}
    for tag, var in (("default", None), ("general", "2")):
        monkeypatch.setenv("AMPLIPY_DEV", "1")
        if var: monkeypatch.setenv("AMPLIPY_KERNEL_VARIANT", var)
        else: monkeypatch.delenv("AMPLIPY_KERNEL_VARIANT", raising=False)
        o = {k: str(tmp_path / ("%s.%s" % (tag, ext))) for k, ext in (("t", "bam"), ("v", "vcf"), ("c", "fas"))}
        amplipy.main(["aio", "-i", inp, "-p", str(bed), "-r", str(ref), "-ot", o["t"], "-ov", o["v"], "-oc", o["c"]])
        outs[tag] = o
    assert open(outs["default"]["v"]).read() == open(outs["general"]["v"]).read()
    assert open(outs["default"]["c"]).read() == open(outs["general"]["c"]).read()
    a, c = bam_native.BamFile(outs["default"]["t"]), bam_native.BamFile(outs["general"]["t"])
    assert a.n_records == c.n_records and a.n_records > 20000      # (AmpliPy.py:910: the reads a primer or quality clip touched)
    ba, _ = a.decode(0, a.n_records, copy=True); bc, _ = c.decode(0, c.n_records, copy=True)
    assert np.array_equal(ba.pos, bc.pos) and np.array_equal(ba.cig, bc.cig) and np.array_equal(ba.cig_off, bc.cig_off)
    a.close(); c.close()


def test_the_kernel_is_chosen_by_the_batch(scheme):
    """amp_last_kernel_variant: 150 bp amplicon reads take the first-generation fast kernel, uniform 250 bp reads the second, a batch
    of mixed long reads with soft clips and indels everywhere (BASELINE config 5) the list-driven one -- each equal to the oracle."""
    from amplipy_amd import lib
    g, pr, amps, mn, mx, mpl = scheme
    e = lib.Engine(g.size)
    e.set_primers(mn, mx, mpl); e.set_params(20, 4, True, True)
    assert e.last_kernel_variant() == 0
    for b, want in ((synth.make_amplicon_batch(g, amps, 30000, seed=5), 4),
                    (synth.make_amplicon_batch(g, amps, 20000, seed=6, read_len=250), 5),
                    (synth.make_config5_batch(g, amps, rep=3, pool_reads=20000, seed=11), 7)):
        e.reset()
        res = e.process(b)
        assert e.last_kernel_variant() == want, (e.last_kernel_variant(), want)
        a = oracle.process(b, g.size, mn, mx, mpl, 20, 4)
        assert np.array_equal(res.new_pos, a.trim.new_pos) and np.array_equal(res.new_ncig, a.trim.new_ncig)
        assert np.array_equal(res.status, a.trim.status) and np.array_equal(e.counts(), a.counts)
    e.close()
