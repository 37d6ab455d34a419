"""Pure-Python restatement of AmpliPy's per-read loop (A:896-915): trim_read (A:426-687) followed by
update_base_counts (A:690-753), one read at a time with Python lists and a per-base interpreter loop --
the algorithmic shape of the reference itself (CPU baseline leg (ii)).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: used by bench.py's cpu_baseline leg (timed on a bounded
sample) and by tests/test_oracle_golden.py, which pins it to the same reference-derived fixtures as the C
restatement.  The product never imports it.  Written from SURVEY.md Appendix A / B, not from the reference's
source: the three clips share one helper here, CIGARs are (op, len) lists, the aligned pairs are generated
lazily instead of being materialised.

process(batch, ref_len, min_start, max_end, max_primer_len, min_quality, window) -> uint32[ref_len][6]
(insertion events are returned by process_full).  Reads outside the reference's domain raise like it does.
"""
from __future__ import annotations

import numpy as np

M, I, D, N, S, H, P, EQ, X = range(9)
CONSUME_QUERY = (1, 1, 0, 0, 1, 0, 0, 1, 1)     # A:43
CONSUME_REF = (1, 0, 1, 1, 0, 0, 0, 1, 1)       # A:44
CODE_COL = {1: 0, 2: 1, 4: 2, 8: 3, 15: 4}      # BAM 4-bit codes of A C G T N -> table column (A:892)


def merge_equal_neighbours(cig):
    """fix_cigar (A:415-423)."""
    out = []
    for op, n in cig:
        if out and out[-1][0] == op:
            out[-1] = (op, out[-1][1] + n)
        else:
            out.append((op, n))
    return out


def pos_on_query(cig, ref_pos, ref_start):
    """A:389-412."""
    q = 0
    cur = ref_start
    for op, n in cig:
        if CONSUME_REF[op]:
            if ref_pos <= cur + n:
                return q + (ref_pos - cur if CONSUME_QUERY[op] else 0)
            cur += n
        if CONSUME_QUERY[op]:
            q += n
    return q


def pos_on_ref(cig, query_pos, ref_start):
    """A:363-386."""
    cur = 0
    r = ref_start
    for op, n in cig:
        if CONSUME_QUERY[op]:
            if query_pos <= cur + n:
                return r + (query_pos - cur if CONSUME_REF[op] else 0)
            cur += n
        if CONSUME_REF[op]:
            r += n
    return r


def reference_length(cig):
    r = sum(n for op, n in cig if op in (M, D, N, EQ, X))
    return r if r else 1


def query_alignment_start(cig):
    off = 0
    for op, n in cig:
        if op == H:
            continue
        if op == S:
            off += n
        else:
            break
    return off


def query_alignment_end(cig, lseq):
    end = lseq
    for k in range(len(cig) - 1, 0, -1):        # element 0 is never examined
        op, n = cig[k]
        if op == H:
            continue
        if op == S:
            end -= n
        else:
            break
    return end


def primer_clip(cig, delete, track):
    """The per-op rules of A:467-510 (start clip; the end clip A:524-555 applies them to the reversed list
    without position tracking).  Returns (new list, reference advance)."""
    out = []
    started = False
    advance = 0
    for op, n in cig:
        if delete == 0 and started:
            out.append((op, n))
            continue
        if delete == 0 and CONSUME_QUERY[op] and CONSUME_REF[op]:
            started = True
            out.append((op, n))
            continue
        ref_add = 0
        if CONSUME_QUERY[op]:
            if delete >= n:
                out.append((S, n))
            elif delete > 0:
                out.append((S, delete))
            else:
                out.append((S, n))
                continue
            ref_add = min(delete, n)
            rest = max(n - delete, 0)
            delete = max(delete - n, 0)
            if rest > 0:
                out.append((op, rest))
            last = out[-1][0]
            if delete == 0 and CONSUME_QUERY[last] and CONSUME_REF[last]:
                started = True
        elif CONSUME_REF[op]:
            ref_add = n
        if track and CONSUME_REF[op]:
            advance += ref_add
    return out, advance


def quality_clip(cig, delete):
    """A:597-622 (and A:658-683 on the reversed list)."""
    out = []
    for op, n in cig:
        if delete == 0 or op in (S, H):
            out.append((op, n))
            continue
        if CONSUME_QUERY[op]:
            out.append((S, min(n, delete)))
            rest = max(n - delete, 0)
            delete = max(delete - n, 0)
            if rest > 0:
                out.append((op, rest))
    return out


def trim_read(cig, pos, flag, tlen, lseq, qual, min_start, max_end, max_primer_len, min_quality, window):
    """A:426-687 -> (new cigar, new pos, flags)."""
    paired, reverse = bool(flag & 1), bool(flag & 0x10)
    left = max_end[pos]
    right = min_start[pos + reference_length(cig) - 1]
    isize = abs(tlen) - max_primer_len > lseq
    f_start = f_end = f_qual = False
    if not (paired and isize and reverse) and left >= 0:
        f_start = True
        delete = pos_on_query(cig, int(left) + 1, pos)
        cig, adv = primer_clip(cig, delete, True)
        cig = merge_equal_neighbours(cig)
        pos += adv
    if not (paired and isize and not reverse) and right >= 0:
        f_end = True
        delete = lseq - pos_on_query(cig, int(right), pos)
        cig, _ = primer_clip(cig[::-1], delete, False)
        cig = merge_equal_neighbours(cig[::-1])
    qs, qe = query_alignment_start(cig), query_alignment_end(cig, lseq)
    q = qual[qs:qe]
    n = len(q)
    w = min(window, n)
    if reverse:
        i = n
        total = sum(int(q[i - off]) for off in range(1, w))
        while i > 0:
            if w > i:
                w -= 1
            else:
                total += int(q[i - w])
            if total < min_quality * w:
                break
            total -= int(q[i - 1])
            i -= 1
        delete = i
        if pos_on_ref(cig, delete + qs - 1, pos) > pos:
            f_qual = True
            cig = merge_equal_neighbours(quality_clip(cig, delete))      # reference_start is not advanced
    else:
        i = 0
        total = sum(int(q[off]) for off in range(w - 1))
        while i < n:
            if n - w < i:
                w -= 1
            else:
                total += int(q[i + w - 1])
            if total < min_quality * w:
                break
            total -= int(q[i])
            i += 1
        delete = n - i
        if delete != 0:
            f_qual = True
            cig = merge_equal_neighbours(quality_clip(cig[::-1], delete)[::-1])
    return cig, pos, (f_start, f_end, f_qual)


def aligned_pairs(cig, pos):
    q, r = 0, pos
    for op, n in cig:
        if op in (M, EQ, X):
            for _ in range(n):
                yield q, r
                q += 1; r += 1
        elif op in (I, S, P):
            for _ in range(n):
                yield q, None
                q += 1
        elif op in (D, N):
            for _ in range(n):
                yield None, r
                r += 1


def update_base_counts(counts, events, cig, pos, lseq, codes, qual, min_quality, read_id):
    """A:690-753.  counts: list of 6-int lists; events: list of (ref_pos, read, q_from, q_to)."""
    qs, qe = query_alignment_start(cig), query_alignment_end(cig, lseq)
    ref_end = pos + reference_length(cig)
    pairs = list(aligned_pairs(cig, pos))
    i = 0
    while i < len(pairs):
        q, r = pairs[i]
        i += 1
        if q is None:
            counts[r][5] += 1
            continue
        if qual[q] < min_quality:
            continue
        if q < qs:
            continue
        if q >= qe:
            break
        if r is None:
            q0 = q
            while r is None and q < qe and qual[q] >= min_quality:
                q, r = pairs[i]
                i += 1
            if r == 0:
                lo, hi = q0, q + 1
            else:
                lo, hi = q0 - 1, (lseq if q is None else q)
            if r is None:
                ins = ref_end
            else:
                ins = r
                i -= 1
            lo2, hi2, _ = slice(lo, hi).indices(lseq)
            events.append((max(ins - 1, 0), read_id, lo2, max(hi2, lo2)))
            continue
        counts[r][CODE_COL[int(codes[q])]] += 1


def process_full(batch, ref_len, min_start, max_end, max_primer_len, min_quality=20, window=4, do_trim=True):
    counts = [[0, 0, 0, 0, 0, 0] for _ in range(ref_len)]
    events = []
    trims = []
    mn, mx = [int(v) for v in min_start], [int(v) for v in max_end]
    for i in range(batch.n):
        c0, c1 = int(batch.cig_off[i]), int(batch.cig_off[i + 1])
        cig = [(int(w) & 15, int(w) >> 4) for w in batch.cig[c0:c1]]
        s0 = int(batch.seq_off[i])
        lseq = int(batch.lseq[i])
        qual = batch.qual[s0:s0 + lseq].tolist()
        packed = batch.seq[s0 // 2:(s0 + lseq + 1) // 2]
        codes = np.empty(packed.size * 2, np.uint8)
        codes[0::2] = packed >> 4
        codes[1::2] = packed & 15
        pos, flags = int(batch.pos[i]), (False, False, False)
        if do_trim:
            cig, pos, flags = trim_read(cig, pos, int(batch.flag[i]), int(batch.tlen[i]), lseq, qual, mn, mx,
                                        max_primer_len, min_quality, window)
        trims.append((pos, cig, flags))
        update_base_counts(counts, events, cig, pos, lseq, codes, qual, min_quality, i)
    return np.array(counts, np.uint32).reshape(ref_len, 6), events, trims


def process(batch, ref_len, min_start, max_end, max_primer_len, min_quality=20, window=4):
    return process_full(batch, ref_len, min_start, max_end, max_primer_len, min_quality, window)[0]
