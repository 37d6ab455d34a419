#!/usr/bin/env python3
"""Times the REFERENCE's own hot-path functions in this container.
Runs only where /root/reference is mounted; imports it by path like tools/make_golden.py."""
import os, sys, time, types
REF_DIR = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isfile(os.path.join(REF_DIR, "AmpliPy.py")):
    sys.exit("reference not mounted")
sys.dont_write_bytecode = True
sys.modules.setdefault("pysam", types.ModuleType("pysam"))
sys.path.insert(0, REF_DIR); sys.path.insert(0, ROOT)
import AmpliPy as REF
from amplipy_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = synth.make_genome(); primers, amps = synth.make_artic_scheme()
pr = sorted((s, e) for s, e, _ in primers)
mn, mx = REF.find_overlapping_primers(g.size, pr, 0); mpl = max(e - s for s, e in pr)
segs = synth.make_amplicon_batch(g, amps, n, seed=1).segments()
counts = [{'A': 0, 'C': 0, 'G': 0, 'T': 0, 'N': 0, '-': 0} for _ in range(g.size)]
t0 = time.perf_counter()
for s in segs:
    REF.trim_read(s, mn, mx, mpl, 20, 4)
t1 = time.perf_counter()
for s in segs:
    REF.update_base_counts(counts, s, 20)
t2 = time.perf_counter()
print("reference AmpliPy.py on %d synthetic 150 bp reads (1 core, Segment stand-in for pysam):" % n)
print("  trim_read          %.1f k reads/s" % (n / (t1 - t0) / 1e3))
print("  update_base_counts %.1f k reads/s" % (n / (t2 - t1) / 1e3))
print("  trim + count       %.1f k reads/s" % (n / (t2 - t0) / 1e3))
