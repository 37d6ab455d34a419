"""Host side of calling (AmpliPy.py:917-952).

The device decides every reference position whose outcome cannot depend on the text of an
insertion allele (``amp_call_positions``).  This module turns those per-position records into
AmpliPy's outputs -- the consensus symbol list and the VCF record fields -- and finishes the
few positions flagged AMP_CALL_INS_RELEVANT from the insertion-event list, where alleles are
strings and ties are broken by Python string order (AmpliPy.py:771).
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np

from . import abi

SYMS = abi.SYMBOLS


class VariantRecord:
    """Fields of one VCF record as AmpliPy assembles them (AmpliPy.py:941-951)."""
    __slots__ = ("pos", "ref", "alts", "DP", "REF_DP", "ALT_DP", "REF_FREQ", "ALT_FREQ", "GT")

    def __init__(self, pos, ref, alts, DP, REF_DP, alt_counts, REF_FREQ, alt_freqs, GT):
        self.pos = pos                      # 0-based; VCF POS = pos + 1 (start=ref_pos, :947)
        self.ref = ref
        self.alts = alts
        self.DP = DP
        self.REF_DP = REF_DP
        self.ALT_DP = ",".join(str(c) for c in alt_counts)          # :944
        self.REF_FREQ = REF_FREQ
        self.ALT_FREQ = ",".join(str(f) for f in alt_freqs)         # :946
        self.GT = GT

    def as_dict(self):
        return {"ref": self.ref, "alts": self.alts, "DP": self.DP, "REF_DP": self.REF_DP, "ALT_DP": self.ALT_DP,
                "REF_FREQ": float(self.REF_FREQ).hex(), "ALT_FREQ": self.ALT_FREQ, "GT": list(self.GT)}


class CallResult:
    def __init__(self, consensus, records, alleles, n_relevant):
        self.consensus = consensus      # list[str | None] per position (None = below thresholds)
        self.records = records          # list[VariantRecord], ascending position
        self.alleles = alleles          # {pos: (total, [(count, freq, symbol)])} when requested
        self.n_relevant = n_relevant


def call_params(min_depth_consensus=10, min_freq_consensus=0.0, min_depth_variants=1, min_freq_variants=0.03,
                run_consensus=True, run_variants=True, full_ranking=False):
    return abi.AmpCallParams(int(min_depth_consensus), int(min_depth_variants), float(min_freq_consensus),
                             float(min_freq_variants), int(bool(run_consensus)), int(bool(run_variants)),
                             int(bool(full_ranking)), 0)


def _ranked_bases(order, counts_row):
    nnz = (int(order) >> 18) & 7
    cols = [(int(order) >> (3 * k)) & 7 for k in range(nnz)]
    return [(int(counts_row[c]), SYMS[c]) for c in cols]


def _decide(pos, ref_symbol, total, ranked, cp):
    """A:928-951 for one position given the ranked (count, freq, symbol) list."""
    consensus = None
    if cp.run_consensus and ranked and ranked[0][0] >= cp.min_depth_consensus and ranked[0][1] >= cp.min_freq_consensus:
        consensus = ranked[0][2]
    record = None
    if cp.run_variants:
        tot = 0; rc = 0; rf = 0; alt_s = []; alt_c = []; alt_f = []
        for c, f, k in ranked:
            tot += c
            if k == ref_symbol:
                rc = c; rf = f
            elif f >= cp.min_freq_variants:
                alt_s.append(k); alt_c.append(c); alt_f.append(f)
        if tot >= cp.min_depth_variants and alt_s:
            gt = tuple(range(len(alt_s) + 1)) if (rc >= cp.min_depth_variants and rf >= cp.min_freq_variants) \
                else tuple(range(1, len(alt_s) + 1))
            record = VariantRecord(pos, ref_symbol, alt_s, total, rc, alt_c, rf, alt_f, gt)
    return consensus, record


def call(engine, ref_seq, cp, ins_strings_at=None, want_alleles=False):
    """Run calling for the state accumulated in ``engine``.

    ``cp``: abi.AmpCallParams (see call_params).  ``ins_strings_at(positions) -> {pos: Counter}``
    supplies the insertion-allele tallies of the flagged positions (only called when needed).
    """
    pc, n_rel = engine.call_positions(cp)
    G = engine.ref_len
    counts = None
    consensus = [None] * G
    if cp.run_consensus:
        cs = pc["consensus_sym"]
        for p in np.nonzero(cs >= 0)[0]:
            consensus[int(p)] = SYMS[int(cs[p])]
    relevant = np.nonzero(pc["flags"] & abi.CALL_INS_RELEVANT)[0]
    rel_set = set(int(p) for p in relevant)
    records = {}
    alleles = {} if want_alleles else None
    var_pos = np.nonzero(pc["flags"] & abi.CALL_VARIANT)[0] if cp.run_variants else []
    need_counts = len(var_pos) or len(relevant) or want_alleles
    if need_counts:
        counts = engine.counts()
    for p in var_pos:
        p = int(p)
        if p in rel_set:
            continue
        rec = pc[p]
        total = int(rec["total_depth"])
        ranked = _ranked_bases(rec["order"], counts[p])
        alt_s = []; alt_c = []; alt_f = []
        for k, (c, s) in enumerate(ranked):
            if (int(rec["alt_mask"]) >> k) & 1:
                alt_s.append(s); alt_c.append(c); alt_f.append(c / total)
        rc = int(rec["ref_count"])
        rf = rc / total if rc else 0
        gt = tuple(range(len(alt_s) + 1)) if (int(rec["flags"]) & abi.CALL_GT_HAS_REF) else tuple(range(1, len(alt_s) + 1))
        records[p] = VariantRecord(p, ref_seq[p], alt_s, total, rc, alt_c, rf, alt_f, gt)
    if len(relevant):
        if ins_strings_at is None:
            raise RuntimeError("%d positions need insertion alleles but no provider was given" % len(relevant))
        tallies = ins_strings_at(rel_set)
        for p in sorted(rel_set):
            d = {SYMS[c]: int(counts[p, c]) for c in range(abi.NSYM)}
            for s, n in tallies.get(p, {}).items():
                d[s] = d.get(s, 0) + n
            total = sum(d.values())
            assert total == int(pc[p]["total_depth"]), "insertion tally does not match the device total at %d" % p
            ranked = sorted(((d[k], d[k] / total, k) for k in d if d[k] != 0), reverse=True)
            cons, rec = _decide(p, ref_seq[p], total, ranked, cp)
            consensus[p] = cons
            records.pop(p, None)
            if rec is not None:
                records[p] = rec
            if want_alleles:
                alleles[p] = (total, ranked)
    if want_alleles:
        nz = np.nonzero(pc["total_depth"])[0]
        for p in nz:
            p = int(p)
            if p in alleles:
                continue
            total = int(pc[p]["total_depth"])
            alleles[p] = (total, [(c, c / total, s) for c, s in _ranked_bases(pc[p]["order"], counts[p])])
    return CallResult(consensus, [records[p] for p in sorted(records)], alleles, n_rel)


def tallies_from_events(events_with_strings, positions):
    """{pos: {string: count}} from [(ref_pos, string)] restricted to ``positions``."""
    out = defaultdict(lambda: defaultdict(int))
    for pos, s in events_with_strings:
        if pos in positions:
            out[pos][s] += 1
    return out
