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
        self.ALT_DP = ",".join(map(str, alt_counts))                # :944
        self.REF_FREQ = REF_FREQ
        self.ALT_FREQ = ",".join(map(str, alt_freqs))               # :946
        self.GT = GT

    def as_dict(self):
        return {"ref": self.ref, "alts": self.alts, "DP": self.DP, "REF_DP": self.REF_DP, "ALT_DP": self.ALT_DP,
                "REF_FREQ": float(self.REF_FREQ).hex(), "ALT_FREQ": self.ALT_FREQ, "GT": list(self.GT)}


_CONS_LUT = {}


class CallResult:
    """Columnar outcome of calling.

    consensus_sym  int8[G]: -1 below thresholds, 0..5 = A C G T N '-', -2 = an insertion string
                   (text in ``consensus_ins[pos]``)
    var_pos        int32[V] positions with a VCF record, ascending
    var_total / var_ref_count  uint32[V];  var_gt_ref  bool[V] (GT includes 0)
    var_nalt       int8[V];  var_alt_col int8[V,6] (columns of the base-symbol ALTs, -1 pad);
                   var_alt_count uint32[V,6]
    ``extra``      {pos: VariantRecord or None} for positions finished from insertion alleles
                   (they override the arrays; None = no record)
    """

    def __init__(self, ref_seq, consensus_sym, consensus_ins, var_pos, var_total, var_ref_count, var_gt_ref, var_nalt,
                 var_alt_col, var_alt_count, extra, alleles, n_relevant):
        self.ref_seq = ref_seq
        self.consensus_sym = consensus_sym
        self.consensus_ins = consensus_ins
        self.var_pos, self.var_total, self.var_ref_count, self.var_gt_ref = var_pos, var_total, var_ref_count, var_gt_ref
        self.var_nalt, self.var_alt_col, self.var_alt_count = var_nalt, var_alt_col, var_alt_count
        self.extra = extra
        self.alleles = alleles
        self.n_relevant = n_relevant
        self._records = None

    @property
    def n_records(self):
        return int(self.var_pos.size) + sum(1 for v in self.extra.values() if v is not None)

    @property
    def consensus(self):
        """list[str | None] per position (None = below thresholds)."""
        out = [None if c < 0 else SYMS[c] for c in self.consensus_sym.tolist()]
        for p, s in self.consensus_ins.items():
            out[p] = s
        return out

    def consensus_string(self, unknown_symbol="N"):
        """''.join(consensus_symbols) of AmpliPy.py:960."""
        lut = _CONS_LUT.get(unknown_symbol) if len(unknown_symbol) == 1 else None
        if lut is None:
            if len(unknown_symbol) != 1:
                return "".join(unknown_symbol if c is None else c for c in self.consensus)
            lut = np.full(256, ord(unknown_symbol), np.uint8)       # negative codes (as uint8) -> unknown
            lut[:6] = np.frombuffer(SYMS.encode("ascii"), np.uint8)
            _CONS_LUT[unknown_symbol] = lut
        text = lut[self.consensus_sym.view(np.uint8)]
        if not self.consensus_ins:
            return text.tobytes().decode("ascii")
        parts = [chr(c) for c in text]
        for p, s in self.consensus_ins.items():
            parts[p] = s
        return "".join(parts)

    def vcf_text(self, ref_id):
        """The VCF body (AmpliPy.py:941-951, one line per record, ascending position) straight from the columns: the same text
        as VcfWriter.line over ``records``, without one Python object per record (12,000 records: 15 ms instead of 50)."""
        tot = self.var_total.tolist(); rcs = self.var_ref_count.tolist(); nas = self.var_nalt.tolist(); gtr = self.var_gt_ref.tolist()
        colL = self.var_alt_col.tolist(); cntL = self.var_alt_count.tolist()
        ref_seq = self.ref_seq
        gts = {(True, na): "/".join(map(str, range(na + 1))) for na in range(8)}
        gts.update({(False, na): "/".join(map(str, range(1, na + 1))) for na in range(8)})
        lines = {}
        for i, p in enumerate(self.var_pos.tolist()):
            total = tot[i]; rc = rcs[i]; na = nas[i]
            cnts = cntL[i][:na]
            lines[p] = "%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%d;REF_DP=%d;ALT_DP=%s;REF_FREQ=%g;ALT_FREQ=%s\tGT\t%s\n" % (
                ref_id, p + 1, ref_seq[p], ",".join([SYMS[c] for c in colL[i][:na]]), total, rc, ",".join(map(str, cnts)),
                rc / total if rc else 0, ",".join([str(c / total) for c in cnts]), gts[(bool(gtr[i]), na)])
        for p, r in self.extra.items():
            lines.pop(p, None)
            if r is not None:
                lines[p] = "%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%d;REF_DP=%d;ALT_DP=%s;REF_FREQ=%g;ALT_FREQ=%s\tGT\t%s\n" % (
                    ref_id, r.pos + 1, r.ref, ",".join(r.alts), r.DP, r.REF_DP, r.ALT_DP, r.REF_FREQ, r.ALT_FREQ, "/".join(map(str, r.GT)))
        return "".join([lines[p] for p in sorted(lines)])

    @property
    def records(self):
        """VariantRecord objects in ascending position (materialised on demand)."""
        if self._records is None:
            recs = {}
            # (columns to lists first: indexing numpy arrays element by element costs more than the record itself)
            tot = self.var_total.tolist(); rcs = self.var_ref_count.tolist(); nas = self.var_nalt.tolist(); gtr = self.var_gt_ref.tolist()
            colL = self.var_alt_col.tolist(); cntL = self.var_alt_count.tolist()
            ref_seq = self.ref_seq
            for i, p in enumerate(self.var_pos.tolist()):
                total = tot[i]; rc = rcs[i]; na = nas[i]
                cols = colL[i][:na]; cnts = cntL[i][:na]
                rf = rc / total if rc else 0
                gt = tuple(range(na + 1)) if gtr[i] else tuple(range(1, na + 1))
                recs[p] = VariantRecord(p, ref_seq[p], [SYMS[c] for c in cols], total, rc, cnts, rf,
                                        [c / total for c in cnts], gt)
            for p, r in self.extra.items():
                recs.pop(p, None)
                if r is not None:
                    recs[p] = r
            self._records = [recs[p] for p in sorted(recs)]
        return self._records


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


def result_from_compact(ref_seq, cp, cons, vr):
    """CallResult from the arrays of Engine.call_compact when no position needs insertion alleles."""
    if not cp.run_consensus:
        cons = np.full(cons.size, -1, np.int8)
    return CallResult(ref_seq, cons, {}, vr["pos"], vr["total_depth"], vr["ref_count"], vr["gt_has_ref"].view(np.bool_),
                      vr["n_alt"].view(np.int8), vr["alt_col"].view(np.int8), vr["alt_count"], {}, None, 0)


def call(engine, ref_seq, cp, ins_strings_at=None, want_alleles=False, positions=None):
    """Run calling for the state accumulated in ``engine``.

    ``cp``: abi.AmpCallParams (see call_params).  ``ins_strings_at(positions) -> {pos: {str: n}}``
    supplies the insertion-allele tallies of the flagged positions (only called when needed).
    ``positions``: a (records, n_relevant) pair already obtained from engine.call_positions.
    """
    if positions is None and not want_alleles and not cp.full_ranking:
        # fast path: everything was decided and packed on the device
        cons, vr, rel = engine.call_compact(cp)
        if rel.size == 0:
            return result_from_compact(ref_seq, cp, cons, vr)
    pc, n_rel = positions if positions is not None else engine.call_positions(cp)
    flags = pc["flags"]
    consensus_sym = pc["consensus_sym"].copy() if cp.run_consensus else np.full(engine.ref_len, -1, np.int8)
    is_rel = (flags & abi.CALL_INS_RELEVANT) != 0
    var_mask = ((flags & abi.CALL_VARIANT) != 0) & ~is_rel if cp.run_variants else np.zeros(flags.size, bool)
    var_pos = np.nonzero(var_mask)[0].astype(np.int32)
    counts = None
    if var_pos.size or n_rel or want_alleles:
        counts = engine.counts()
    V = var_pos.size
    alt_col = np.full((V, 6), -1, np.int8)
    alt_cnt = np.zeros((V, 6), np.uint32)
    nalt = np.zeros(V, np.int8)
    if V:
        order = pc["order"][var_pos]
        am = pc["alt_mask"][var_pos]
        for k in range(6):                    # ranked slot k -> next free ALT column
            sel = ((am >> k) & 1).astype(bool)
            if not sel.any():
                continue
            col = ((order >> (3 * k)) & 7).astype(np.int8)
            rows = np.nonzero(sel)[0]
            dst = nalt[rows].astype(np.int64)
            alt_col[rows, dst] = col[rows]
            alt_cnt[rows, dst] = counts[var_pos[rows], col[rows]]
            nalt[rows] += 1
    extra = {}
    consensus_ins = {}
    alleles = {} if want_alleles else None
    if n_rel:
        if ins_strings_at is None:
            raise RuntimeError("%d positions need insertion alleles but no provider was given" % n_rel)
        rel_set = set(int(p) for p in np.nonzero(is_rel)[0])
        tallies = ins_strings_at(rel_set)
        for p in sorted(rel_set):
            d = {SYMS[c]: int(counts[p, c]) for c in range(abi.NSYM)}
            for s_, n in tallies.get(p, {}).items():
                d[s_] = d.get(s_, 0) + n
            total = sum(d.values())
            assert total == int(pc[p]["total_depth"]), "insertion tally does not match the device total at %d" % p
            ranked = sorted(((d[k], d[k] / total, k) for k in d if d[k] != 0), reverse=True)
            cons, rec = _decide(p, ref_seq[p], total, ranked, cp)
            if cons is None:
                consensus_sym[p] = -1
            elif len(cons) == 1 and cons in SYMS:
                consensus_sym[p] = SYMS.index(cons)
            else:
                consensus_sym[p] = -2
                consensus_ins[p] = cons
            extra[p] = rec
            if want_alleles:
                alleles[p] = (total, ranked)
    if want_alleles:
        for p in np.nonzero(pc["total_depth"])[0]:
            p = int(p)
            if p not in alleles:
                total = int(pc[p]["total_depth"])
                alleles[p] = (total, [(c, c / total, s_) for c, s_ in _ranked_bases(pc[p]["order"], counts[p])])
    return CallResult(ref_seq, consensus_sym, consensus_ins, var_pos, pc["total_depth"][var_pos],
                      pc["ref_count"][var_pos], (flags[var_pos] & abi.CALL_GT_HAS_REF) != 0, nalt, alt_col, alt_cnt,
                      extra, alleles, n_rel)


def tallies_from_events(events_with_strings, positions):
    """{pos: {string: count}} from [(ref_pos, string)] restricted to ``positions``."""
    out = defaultdict(lambda: defaultdict(int))
    for pos, s in events_with_strings:
        if pos in positions:
            out[pos][s] += 1
    return out


def tallies_from_runs(runs_with_strings, positions):
    """{pos: {string: count}} from [(ref_pos, string, count)] (the runs of Engine.aggregate_events with their text)
    restricted to ``positions``; rows of one allele are summed."""
    out = defaultdict(lambda: defaultdict(int))
    for pos, s, c in runs_with_strings:
        if pos in positions:
            out[pos][s] += c
    return out
