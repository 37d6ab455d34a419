"""Shared helpers: golden-fixture loading and comparison of engine output with expectations."""
import gzip
import json
import os
from collections import Counter

import numpy as np

from amplipy_amd import abi
from amplipy_amd.batch import ReadBatch
from amplipy_amd.insertions import event_strings
from amplipy_amd.segment import Segment

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SYM_COL = {s: i for i, s in enumerate(abi.SYMBOLS)}


def load_json(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        with gzip.open(path, "rt") as f:
            return json.load(f)
    with open(path) as f:
        return json.load(f)


def seg_from_dict(d):
    return Segment(flag=d["flag"], reference_start=d["pos"], cigar=d["cigar"], template_length=d["tlen"],
                   query_sequence=d["seq"], query_qualities=d["qual"])


def sparse_from_engine(counts, batch, events, read_base=0):
    """Engine state -> Counter {(pos, symbol_string): n}, the shape of the golden 'counts' lists."""
    out = Counter()
    nz = np.argwhere(counts)
    for p, c in nz:
        out[(int(p), abi.SYMBOLS[c])] += int(counts[p, c])
    for pos, s in event_strings(batch, events, read_base):
        out[(pos, s)] += 1
    return out


def sparse_from_golden(lst):
    return Counter({(int(p), k): int(n) for p, k, n in lst})


def exc_name(status):
    e = abi.READ_STATUS_EXC[int(status)]
    return None if e is None else e.__name__


def check_case_with(process_fn, case, tables_fn):
    """Run every read of a golden case through ``process_fn`` one read per batch and compare with
    the reference outcome.  process_fn(batch, ref_len, mn, mx, mpl, min_quality, window, do_trim)
    -> object with .trim (abi.TrimResult), .counts, .events."""
    G = case["ref_len"]
    mn, mx, mpl = tables_fn(G, case["primers"], case["offset"])
    assert mpl == case["max_primer_len"]
    mq, w = case["min_quality"], case["window"]
    fails = []
    for idx, (rd, exp) in enumerate(zip(case["reads"], case["expected"])):
        b = ReadBatch.from_segments([seg_from_dict(rd)])
        tag = "%s[%d] %s" % (case["name"], idx, rd["cigar"])
        r = process_fn(b, G, mn, mx, mpl, mq, w, True)
        t = exp["trim"]
        st = int(r.trim.status[0])
        if "error" in t:
            if exc_name(st) != t["error"]:
                fails.append("%s: trim expected %s got status %d" % (tag, t["error"], st))
        else:
            ct = exp["count_trimmed"]
            got = (int(r.trim.new_pos[0]), r.trim.cigar_string(0),
                   [bool(r.trim.trim_flags[0] & 1), bool(r.trim.trim_flags[0] & 2), bool(r.trim.trim_flags[0] & 4)],
                   int(r.trim.ref_len[0]))
            want = (t["pos"], t["cigar"], t["flags"], t["reflen"])
            if "error" in ct:
                if exc_name(st) != ct["error"]:
                    fails.append("%s: count expected %s got status %d" % (tag, ct["error"], st))
            else:
                if st != 0:
                    fails.append("%s: unexpected status %d" % (tag, st))
                elif got != want:
                    fails.append("%s: trim %r != %r" % (tag, got, want))
                elif sparse_from_engine(r.counts, b, r.events) != sparse_from_golden(ct["counts"]):
                    fails.append("%s: counts differ" % tag)
        # counting without trimming (variants / consensus modes)
        r = process_fn(b, G, mn, mx, mpl, mq, w, False)
        cr = exp["count_raw"]
        st = int(r.trim.status[0])
        if "error" in cr:
            if exc_name(st) != cr["error"]:
                fails.append("%s: raw count expected %s got status %d" % (tag, cr["error"], st))
        elif st != 0:
            fails.append("%s: raw count unexpected status %d" % (tag, st))
        elif sparse_from_engine(r.counts, b, r.events) != sparse_from_golden(cr["counts"]):
            fails.append("%s: raw counts differ" % tag)
    return fails
