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


class EventStore:
    """Insertion events of many batches with their allele text, kept columnar: positions, one byte
    blob and offsets.  Adding a batch is vectorised (no per-event Python); strings are only built for
    the positions calling asks about (AmpliPy.py:745-748 keys, restricted to where they can matter)."""

    def __init__(self):
        self._pos = []; self._len = []; self._blob = []; self._cnt = []

    def add(self, batch, events, read_base=0):
        if events.size == 0:
            return
        i = (events["read"].astype(np.int64) - int(read_base)) & 0xFFFFFFFF      # read ids are 32-bit: relative to read_base modulo 2^32
        start = batch.seq_off[i].astype(np.int64) + events["q_from"].astype(np.int64)
        length = np.maximum(events["q_to"].astype(np.int64) - events["q_from"].astype(np.int64), 0)
        total = int(length.sum())
        first = np.cumsum(length) - length                       # offset of each event in the blob
        b = np.repeat(start - first, length) + np.arange(total, dtype=np.int64)     # base index of every blob byte
        nib = (batch.seq[b >> 1] >> ((1 - (b & 1)) * 4).astype(np.uint8)) & 15
        self._pos.append(events["ref_pos"].astype(np.int64)); self._len.append(length); self._blob.append(_NT16[nib]); self._cnt.append(np.ones(length.size, np.int64))

    def add_text(self, ref_pos, length, blob, count=None):
        """The same from allele text gathered elsewhere (Engine.event_text: on the device, from the batch it just processed);
        ``count`` = events per row when the rows are the runs of Engine.aggregate_events (one row per allele), else one each."""
        if len(ref_pos):
            self._pos.append(np.asarray(ref_pos, np.int64)); self._len.append(np.asarray(length, np.int64)); self._blob.append(np.asarray(blob, np.uint8))
            self._cnt.append(np.ones(len(ref_pos), np.int64) if count is None else np.asarray(count, np.int64))

    def __len__(self):
        return int(sum(int(c.sum()) for c in self._cnt))

    def counted_pairs(self, positions=None):
        """[(ref_pos, string, count)] of the rows at ``positions`` (all rows when None)."""
        out = []
        want = None if positions is None else np.fromiter(positions, np.int64, len(positions))
        for pos, length, blob, cnt in zip(self._pos, self._len, self._blob, self._cnt):
            off = np.cumsum(length) - length
            sel = np.arange(pos.size) if want is None else np.nonzero(np.isin(pos, want))[0]
            raw = blob.tobytes()
            for k in sel.tolist():
                o = int(off[k])
                out.append((int(pos[k]), raw[o:o + int(length[k])].decode("ascii"), int(cnt[k])))
        return out

    def pairs(self, positions=None):
        """[(ref_pos, string)] of the events at ``positions`` (all events when None), in insertion order."""
        return [(p, s) for p, s, c in self.counted_pairs(positions) for _ in range(c)]


def tally(pairs):
    """Counter {(ref_pos, string): count}."""
    return Counter(pairs)
