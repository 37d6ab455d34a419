"""Host side of insertion alleles (AmpliPy.py:730-748).

The device records integer events ``(ref_pos, read, q_from, q_to)``; the allele itself is the
upper-cased read substring ``SEQ[q_from:q_to]`` (AmpliPy.py:702, :736-738), materialised here
and tallied per ``(ref_pos, string)`` -- the dynamic dict keys of AmpliPy.py:745-748.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

from .batch import SEQ_NT16

_NT16 = np.frombuffer(SEQ_NT16.encode("ascii"), dtype=np.uint8)


def event_strings(batch, events, read_base=0):
    """[(ref_pos, string)] for every event, in event order."""
    out = []
    for ev in events:
        i = int(ev["read"]) - read_base
        lo = int(batch.seq_off[i]) + int(ev["q_from"])
        hi = int(batch.seq_off[i]) + int(ev["q_to"])
        if hi <= lo:
            out.append((int(ev["ref_pos"]), ""))
            continue
        packed = batch.seq[lo // 2:(hi + 1) // 2]
        codes = np.empty(packed.size * 2, np.uint8)
        codes[0::2] = packed >> 4
        codes[1::2] = packed & 15
        codes = codes[lo - 2 * (lo // 2): lo - 2 * (lo // 2) + (hi - lo)]
        out.append((int(ev["ref_pos"]), _NT16[codes].tobytes().decode("ascii")))
    return out


def tally(pairs):
    """Counter {(ref_pos, string): count}."""
    return Counter(pairs)
