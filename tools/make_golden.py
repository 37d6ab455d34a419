#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own functions in this container.

Run as ``python3 -B tools/make_golden.py`` from the repo root (``--check``: regenerate into a scratch
directory and compare with the committed fixtures; tests/test_golden_generator.py runs that).  The script locates
``/root/reference/AmpliPy.py`` at run time and refuses to run when it is absent (it is absent
on the GPU box; only the generated fixtures travel) and loads it from that file, never by module
name.  The reference's import of ``pysam`` is needed only
for file I/O, which the hot path never touches, so an empty module object is registered under
that name before the import.  The reference's ``trim_read`` /
``update_base_counts`` are then called UNMODIFIED on ``amplipy_amd.segment.Segment`` records,
which implement the pysam accessor semantics of SURVEY.md Appendix B.

What the fixtures pin: find_overlapping_primers, get_pos_on_query, get_pos_on_ref, fix_cigar,
trim_read, update_base_counts, alleles_from_counts (all the reference's code).  What they do
not pin: pysam's derived properties (Appendix B, restated by Segment) and the calling loop
AmpliPy.py:921-951, which cannot run without pysam and is restated in ``call_positions``
below around the imported ``alleles_from_counts`` (marked ``"calls_restated": true``).

Only data is written: inputs and the reference's outputs.  No reference source is copied.
"""
import gzip
import importlib.util
import json
import os
import shutil
import sys
import tempfile
import types

REF_DIR = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

if not os.path.isfile(os.path.join(REF_DIR, "AmpliPy.py")):
    sys.exit("make_golden: %s/AmpliPy.py not found -- fixtures can only be generated where the "
             "reference is mounted" % REF_DIR)

sys.dont_write_bytecode = True


def load_reference():
    """The reference module, loaded from its FILE -- never by module name: the repo root holds an
    `AmpliPy.py` of its own (the drop-in entry point), and fixtures pinned to that would pin the oracle
    to the product.  The origin is asserted."""
    sys.modules.setdefault("pysam", types.ModuleType("pysam"))
    path = os.path.join(REF_DIR, "AmpliPy.py")
    spec = importlib.util.spec_from_file_location("amplipy_reference", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    origin = os.path.realpath(mod.__file__)
    assert origin.startswith(os.path.realpath(REF_DIR) + os.sep), "reference loaded from %s" % origin
    for name in ("find_overlapping_primers", "get_pos_on_query", "get_pos_on_ref", "fix_cigar", "trim_read",
                 "update_base_counts", "alleles_from_counts", "load_primers", "load_ref_genome", "VERSION"):
        assert hasattr(mod, name), "reference module lacks %s" % name
    return mod


REF = load_reference()
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from amplipy_amd.segment import Segment, format_cigar, parse_cigar  # noqa: E402
from amplipy_amd import synth  # noqa: E402

SYMS = "ACGTN-"


class LazyTable:
    """list-of-dicts stand-in for symbol_counts_at_ref_pos (AmpliPy.py:892) that only
    materialises touched positions; indexing follows list semantics (IndexError past the
    end, negative indices wrap)."""

    def __init__(self, n):
        self.n = n
        self.d = {}

    def __getitem__(self, i):
        if i is None or not isinstance(i, int):
            raise TypeError("list indices must be integers")
        if i >= self.n or i < -self.n:
            raise IndexError("list index out of range")
        if i < 0:
            i += self.n
        if i not in self.d:
            self.d[i] = {"A": 0, "C": 0, "G": 0, "T": 0, "N": 0, "-": 0}
        return self.d[i]

    def sparse(self):
        out = []
        for p in sorted(self.d):
            for k in sorted(self.d[p]):
                if self.d[p][k]:
                    out.append([p, k, self.d[p][k]])
        return out


def none_to_neg(lst):
    return [-1 if v is None else int(v) for v in lst]


def read_dict(s):
    return {"flag": s.flag, "pos": s.reference_start, "cigar": s.cigarstring,
            "tlen": s.template_length, "seq": s.query_sequence,
            "qual": None if s.query_qualities is None else
            "".join(chr(q + 33) for q in s.query_qualities)}


def seg_from_dict(d):
    return Segment(flag=d["flag"], reference_start=d["pos"], cigar=d["cigar"],
                   template_length=d["tlen"], query_sequence=d["seq"], query_qualities=d["qual"])


def run_read(d, ref_len, min_start, max_end, max_primer_len, min_quality, window):
    """Reference outcome for one read: trim, count-after-trim, count-without-trim."""
    out = {}
    s = seg_from_dict(d)
    try:
        flags = REF.trim_read(s, min_start, max_end, max_primer_len, min_quality, window)
        out["trim"] = {"pos": s.reference_start, "cigar": s.cigarstring,
                       "flags": [bool(f) for f in flags], "reflen": s.reference_length}
    except Exception as e:  # the reference raises on out-of-domain inputs
        out["trim"] = {"error": type(e).__name__}
    if "error" not in out["trim"]:
        t = LazyTable(ref_len)
        try:
            REF.update_base_counts(t, s, min_quality)
            out["count_trimmed"] = {"counts": t.sparse()}
        except Exception as e:
            out["count_trimmed"] = {"error": type(e).__name__}
    s = seg_from_dict(d)
    t = LazyTable(ref_len)
    try:
        REF.update_base_counts(t, s, min_quality)
        out["count_raw"] = {"counts": t.sparse()}
    except Exception as e:
        out["count_raw"] = {"error": type(e).__name__}
    return out


def primer_tables(ref_len, primers, offset):
    ps = sorted((int(a), int(b)) for a, b in primers)
    mn, mx = REF.find_overlapping_primers(ref_len, ps, offset)
    return mn, mx, max(e - s for s, e in ps)


def case(name, ref_len, primers, offset, min_quality, window, reads, note=None):
    mn, mx, mpl = primer_tables(ref_len, primers, offset)
    res = [run_read(d, ref_len, mn, mx, mpl, min_quality, window) for d in reads]
    c = {"name": name, "ref_len": ref_len, "primers": [[int(a), int(b)] for a, b in primers],
         "offset": offset, "min_quality": min_quality, "window": window,
         "max_primer_len": mpl, "reads": reads, "expected": res}
    if note:
        c["note"] = note
    return c


def call_positions(ref_seq, table, params):
    """AmpliPy.py:921-951 restated around the reference's alleles_from_counts."""
    out = []
    for p in sorted(table.d):
        total, alleles = REF.alleles_from_counts(table.d[p])
        rec = {"pos": p, "total": total,
               "alleles": [[c, float(f).hex(), k] for c, f, k in alleles]}
        if alleles and alleles[0][0] >= params["min_depth_consensus"] and \
                alleles[0][1] >= params["min_freq_consensus"]:
            rec["consensus"] = alleles[0][2]
        ref_symbol = ref_seq[p]
        tot = 0; rc = 0; rf = 0; alt_s = []; alt_c = []; alt_f = []
        for c, f, k in alleles:
            tot += c
            if k == ref_symbol:
                rc = c; rf = f
            elif f >= params["min_freq_variants"]:
                alt_s.append(k); alt_c.append(c); alt_f.append(f)
        if tot >= params["min_depth_variants"] and alt_s:
            gt = list(range(len(alt_s) + 1)) if (rc >= params["min_depth_variants"] and
                                                 rf >= params["min_freq_variants"]) \
                else list(range(1, len(alt_s) + 1))
            rec["variant"] = {"ref": ref_symbol, "alts": alt_s, "DP": total, "REF_DP": rc,
                              "ALT_DP": ",".join(str(c) for c in alt_c),
                              "REF_FREQ": float(rf).hex(),
                              "ALT_FREQ": ",".join(str(f) for f in alt_f), "GT": gt}
        out.append(rec)
    return out


def dump(name, obj, gz=False):
    path = os.path.join(OUT, name)
    data = json.dumps(obj, separators=(",", ":"))
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data.encode())
    else:
        with open(path, "w") as f:
            f.write(data)
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))


def R(flag, pos, cigar, tlen=0, seq=None, qual=None, q=None, rng=None):
    """Build a read dict; seq defaults to a deterministic ACGT string, qual to all-F."""
    ops = parse_cigar(cigar)
    L = sum(n for o, n in ops if o in (0, 1, 4, 7, 8))
    if seq is None:
        seq = "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(L))
    if qual is None:
        qual = [37] * L if q is None else list(q)
    if not isinstance(qual, str):
        qual = "".join(chr(v + 33) for v in qual)
    return {"flag": flag, "pos": pos, "cigar": cigar, "tlen": tlen, "seq": seq, "qual": qual}


def lowq(L, lo, hi, val=2, base=37):
    v = [base] * L
    for i in range(lo, hi):
        v[i] = val
    return v


def main():
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    meta = {"reference": "Niema-Lab/AmpliPy", "reference_version": REF.VERSION,
            "generator": "tools/make_golden.py"}

    # ---- example data files (data, not source) --------------------------------------
    for fn in ("example_primers.bed", "example_reference.fas",
               "example_primer_trim_start.sam", "example_primer_trim_end.sam"):
        shutil.copyfile(os.path.join(REF_DIR, "example", fn), os.path.join(OUT, "data", fn))
    ex_primers = REF.load_primers(os.path.join(REF_DIR, "example", "example_primers.bed"))
    ex_id, ex_ref = REF.load_ref_genome(os.path.join(REF_DIR, "example", "example_reference.fas"))
    G = len(ex_ref)

    # ---- 1. primer tables -------------------------------------------------------------
    tabs = {}
    for off in (0, 5):
        mn, mx, mpl = primer_tables(G, ex_primers, off)
        tabs["example_off%d_min_start" % off] = np.array(none_to_neg(mn), np.int32)
        tabs["example_off%d_max_end" % off] = np.array(none_to_neg(mx), np.int32)
    rng = np.random.default_rng(11)
    sets = []
    for k in range(40):
        glen = int(rng.integers(50, 2000))
        npr = int(rng.integers(1, 40))
        ps = []
        for _ in range(npr):
            a = int(rng.integers(0, glen)); ps.append((a, a + int(rng.integers(1, 40))))
        off = int(rng.integers(0, 8))
        mn, mx, mpl = primer_tables(glen, ps, off)
        tabs["rand%d_min_start" % k] = np.array(none_to_neg(mn), np.int32)
        tabs["rand%d_max_end" % k] = np.array(none_to_neg(mx), np.int32)
        sets.append({"ref_len": glen, "primers": [list(p) for p in ps], "offset": off,
                     "max_primer_len": mpl})
    np.savez_compressed(os.path.join(OUT, "primer_tables.npz"), **tabs)
    dump("primer_tables.json", {"meta": meta, "example": {"ref_len": G, "ref_id": ex_id,
                                                          "n_primers": len(ex_primers),
                                                          "max_primer_len": 30},
                                "random_sets": sets})

    # ---- 2. coordinate helpers ----------------------------------------------------------
    helper = []
    rng = np.random.default_rng(12)
    for _ in range(1500):
        nops = int(rng.integers(1, 8))
        cig = [(int(rng.integers(0, 9)), int(rng.integers(0, 30))) for _ in range(nops)]
        start = int(rng.integers(0, 500))
        x = int(rng.integers(-5, 200))
        helper.append({"cigar": [list(c) for c in cig], "start": start, "x": x,
                       "pos_on_query": REF.get_pos_on_query(cig, start + x, start),
                       "pos_on_ref": REF.get_pos_on_ref(cig, x, start),
                       "fix_cigar": [list(c) for c in REF.fix_cigar(list(cig))]})
    dump("helpers.json", {"meta": meta, "cases": helper})

    # ---- 3. the two example reads ---------------------------------------------------------
    ex_reads = []
    for fn in ("example_primer_trim_start.sam", "example_primer_trim_end.sam"):
        for line in open(os.path.join(REF_DIR, "example", fn)):
            if line.startswith("@"):
                continue
            f = line.rstrip("\n").split("\t")
            ex_reads.append({"flag": int(f[1]), "pos": int(f[3]) - 1, "cigar": f[5],
                             "tlen": int(f[8]), "seq": f[9], "qual": f[10]})
    cases = [case("example_reads", G, ex_primers, 0, 20, 4, ex_reads)]

    # ---- 4. named Appendix-A cases ----------------------------------------------------------
    P1 = [(100, 120)]
    Gs = 1000
    named = [
        ("start_clip_deletion_inside", P1, 0, [R(0, 100, "15M3D85M")]),
        ("start_clip_deletion_after", P1, 0, [R(0, 100, "21M3D79M")]),
        ("start_clip_insertion_after", P1, 0, [R(0, 100, "21M3I76M")]),
        ("leading_hard_clip_dropped", P1, 0, [R(0, 100, "5H10S90M")]),
        ("read_inside_primer", P1, 0, [R(0, 105, "10M")]),
        ("offset5_negative_delete", P1, 5, [R(0, 123, "100M")]),
        ("qual_fwd_tail", P1, 0, [R(0, 300, "100M", q=lowq(100, 90, 100))]),
        ("qual_rev_head10", P1, 0, [R(16, 300, "100M", q=lowq(100, 0, 10))]),
        ("qual_rev_head1", P1, 0, [R(16, 300, "100M", q=lowq(100, 0, 1))]),
        ("qual_all_low_fwd", P1, 0, [R(0, 300, "50M", q=[2] * 50)]),
        ("qual_all_low_rev", P1, 0, [R(16, 300, "50M", q=[2] * 50)]),
        ("insertion_mid", P1, 0, [R(0, 200, "50M4I46M")]),
        ("insertion_at_ref0", P1, 0, [R(0, 0, "5S3I92M")]),
        ("insertion_lowq_third", P1, 0, [R(0, 200, "50M4I46M", q=lowq(100, 52, 53))]),
        ("insertion_before_trailing_clip", P1, 0, [R(0, 200, "90M4I6S")]),
        ("insertion_then_deletion", P1, 0, [R(0, 200, "50M4I3D46M")]),
        ("insertion_first_op", P1, 0, [R(0, 200, "2I98M")]),
        ("insertion_first_base_lowq", P1, 0, [R(0, 200, "50M4I46M", q=lowq(100, 50, 51))]),
        ("cigar_ends_in_insertion", P1, 0, [R(0, 200, "98M2I")]),
        ("cigar_ends_in_insertion_lowq", P1, 0, [R(0, 200, "98M2I", q=lowq(100, 99, 100))]),
        ("iupac_base", P1, 0, [R(0, 200, "10M", seq="ACGTRACGTA")]),
        ("lowercase_bases", P1, 0, [R(0, 200, "10M", seq="acgtnACGTN")]),
        ("past_reference_end", P1, 0, [R(0, 950, "100M")]),
        ("touches_reference_end", P1, 0, [R(0, 900, "100M")]),
        ("ref_skip_and_pad", P1, 0, [R(0, 200, "20M5N20M2P20M")]),
        ("eq_and_diff_ops", P1, 0, [R(0, 95, "20=5X75=")]),
        ("paired_isize_rev_skips_start", P1, 0, [R(83, 100, "100M", tlen=-400)]),
        ("paired_isize_fwd_skips_end", [(180, 200)], 0, [R(99, 95, "100M", tlen=400)]),
        ("paired_small_isize_both", [(100, 120), (180, 200)], 0, [R(99, 100, "100M", tlen=120)]),
        ("end_clip_with_trailing_hard", [(180, 200)], 0, [R(0, 95, "100M20H")]),
        ("end_clip_deletion_in_tail", [(180, 200)], 0, [R(0, 95, "80M3D20M")]),
        ("end_clip_insertion_in_tail", [(180, 200)], 0, [R(0, 95, "90M2I8M")]),
        ("start_and_end_and_quality", [(100, 120), (180, 200)], 0,
         [R(0, 100, "100M", q=lowq(100, 60, 70))]),
        ("window_longer_than_read", P1, 0, [R(0, 300, "3M", q=[30, 10, 30])]),
        ("zero_length_alignment", P1, 0, [R(0, 300, "10S")]),
        ("soft_then_hard_lead", P1, 0, [R(0, 300, "5S3H20M")]),
        ("deletion_before_trailing_clip", P1, 0, [R(0, 300, "20M3D5S")]),
        ("insertion_adjacent_to_leading_clip", P1, 0, [R(0, 300, "5S3I20M")]),
        ("qual_fwd_trim_into_insertion", P1, 0, [R(0, 300, "40M5I5M", q=lowq(50, 42, 50))]),
        ("qual_fwd_trim_over_deletion", P1, 0, [R(0, 300, "40M5D10M", q=lowq(50, 35, 50))]),
        ("qual_rev_trim_over_deletion", P1, 0, [R(16, 300, "10M5D40M", q=lowq(50, 0, 15))]),
        ("qual_rev_trim_with_lead_clip", P1, 0, [R(16, 300, "5S45M", q=lowq(50, 5, 12))]),
    ]
    for name, pr, off, reads in named:
        cases.append(case(name, Gs, pr, off, 20, 4, reads))
    cases.append(case("window1", Gs, P1, 0, 20, 1, [R(0, 300, "50M", q=lowq(50, 30, 31)),
                                                     R(16, 300, "50M", q=lowq(50, 30, 31))]))
    cases.append(case("window10_q30", Gs, P1, 0, 30, 10, [R(0, 300, "50M", q=lowq(50, 30, 36, val=5)),
                                                          R(16, 300, "50M", q=lowq(50, 10, 16, val=5))]))
    cases.append(case("min_quality0", Gs, P1, 0, 0, 4, [R(0, 100, "100M", q=lowq(100, 90, 100, val=0))]))
    dump("named_cases.json", {"meta": meta, "cases": cases})

    # ---- 5. seeded random reads over several parameter sets -----------------------------------
    syn_genome = synth.make_genome()
    syn_primers, syn_amps = synth.make_artic_scheme()
    syn_pr2 = [(s, e) for s, e, _ in syn_primers]
    rcases = []
    combos = [("example_bed", G, ex_primers, 0, 20, 4), ("example_bed_off5", G, ex_primers, 5, 20, 4),
              ("synthetic_bed", syn_genome.size, syn_pr2, 0, 20, 4),
              ("synthetic_bed_w1_q0", syn_genome.size, syn_pr2, 0, 0, 1),
              ("synthetic_bed_w10_q30", syn_genome.size, syn_pr2, 0, 30, 10),
              ("example_bed_w7_q25", G, ex_primers, 2, 25, 7)]
    for ci, (nm, glen, pr, off, mq, w) in enumerate(combos):
        rng = np.random.default_rng(100 + ci)
        segs = synth.random_segments(rng, 400, glen, pr)
        rcases.append(case("random_" + nm, glen, pr, off, mq, w, [read_dict(s) for s in segs]))
    dump("random_reads.json.gz", {"meta": meta, "cases": rcases}, gz=True)

    # ---- 6. a 5,000-read synthetic pileup with calls -------------------------------------------
    batch = synth.make_amplicon_batch(syn_genome, syn_amps, 4000, seed=5)
    segs = batch.segments() + synth.make_mixed_segments(syn_genome, syn_amps, 1000, seed=6)
    segs.sort(key=lambda s: s.reference_start)
    mn, mx, mpl = primer_tables(syn_genome.size, syn_pr2, 0)
    params = {"min_quality": 20, "window": 4, "min_depth_consensus": 10, "min_freq_consensus": 0,
              "min_depth_variants": 1, "min_freq_variants": 0.03, "min_length": 30}
    table = LazyTable(syn_genome.size)
    trim_out = []
    reads = [read_dict(s) for s in segs]
    for s in segs:
        fl = REF.trim_read(s, mn, mx, mpl, params["min_quality"], params["window"])
        trim_out.append([s.reference_start, s.cigarstring, [bool(f) for f in fl], s.reference_length])
        REF.update_base_counts(table, s, params["min_quality"])
    ref_str = synth.genome_string(syn_genome)
    dump("pileup_5000.json.gz",
         {"meta": meta, "calls_restated": True, "ref_len": syn_genome.size, "ref_seq": ref_str,
          "primers": [list(p) for p in syn_pr2], "offset": 0, "params": params,
          "max_primer_len": mpl, "reads": reads, "trim": trim_out, "counts": table.sparse(),
          "calls": call_positions(ref_str, table, params)}, gz=True)

    # ---- 7. a second pileup without trimming (variants/consensus modes) ----------------------
    table = LazyTable(syn_genome.size)
    for d in reads[:1500]:
        REF.update_base_counts(table, seg_from_dict(d), 20)
    p2 = dict(params, min_depth_consensus=3, min_freq_consensus=0.6, min_depth_variants=4,
              min_freq_variants=0.1)
    dump("pileup_notrim_1500.json.gz",
         {"meta": meta, "calls_restated": True, "ref_len": syn_genome.size, "n_reads": 1500,
          "params": p2, "counts": table.sparse(), "calls": call_positions(ref_str, table, p2)},
         gz=True)


def check():
    """Regenerate everything into a scratch directory and compare with tests/golden byte for byte
    (.npz members array for array: the zip container carries timestamps)."""
    global OUT
    committed = OUT
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        OUT = tmp
        try:
            main()
        finally:
            OUT = committed
        names = sorted(f for f in os.listdir(tmp) if os.path.isfile(os.path.join(tmp, f)))
        names += sorted(os.path.join("data", f) for f in os.listdir(os.path.join(tmp, "data")))
        for fn in names:
            a, b = os.path.join(tmp, fn), os.path.join(committed, fn)
            if not os.path.isfile(b):
                bad.append(fn + " (not committed)")
            elif fn.endswith(".npz"):
                za, zb = np.load(a), np.load(b)
                if sorted(za.files) != sorted(zb.files) or any(not np.array_equal(za[k], zb[k]) for k in za.files):
                    bad.append(fn)
            elif open(a, "rb").read() != open(b, "rb").read():
                bad.append(fn)
        extra = sorted(set(f for f in os.listdir(committed) if os.path.isfile(os.path.join(committed, f))) - set(names))
        bad += [fn + " (committed, not generated)" for fn in extra]
    if bad:
        sys.exit("make_golden --check: DIFFERENT from tests/golden: " + ", ".join(bad))
    print("make_golden --check: %d files identical to tests/golden (reference %s from %s)"
          % (len(names), REF.VERSION, REF.__file__))


if __name__ == "__main__":
    if "--check" in sys.argv[1:]:
        check()
    else:
        main()
