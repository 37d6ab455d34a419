"""Synthetic workloads for tests and benchmarks.

Nothing here comes from the reference: AmpliPy ships no generators.  The shapes
follow BASELINE.json's configs: a 29,903 nt genome, an ARTIC-v3-style primer
scheme (98 amplicons, two pools, 22-30 nt primers) and 150 bp paired reads that
start at a forward primer or end at a reverse primer.
"""
from __future__ import annotations

import numpy as np

from .batch import ReadBatch, SEQ_NT16, pack_nibbles, ALIGN
from .segment import Segment, OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X

GENOME_LEN = 29903
GENOME_SEED = 20261003
ACGT_CODES = np.array([1, 2, 4, 8], dtype=np.uint8)  # A C G T in BAM 4-bit codes
QUAL_VALUES = np.array([37, 25, 11, 2], dtype=np.uint8)
QUAL_PROBS = np.array([0.80, 0.12, 0.06, 0.02])


def make_genome(length=GENOME_LEN, seed=GENOME_SEED):
    """Uniform ACGT genome as 4-bit codes (uint8[length])."""
    rng = np.random.default_rng(seed)
    return ACGT_CODES[rng.integers(0, 4, size=length)]


def genome_string(codes):
    return "".join(SEQ_NT16[c] for c in codes)


def make_artic_scheme(length=GENOME_LEN, n_amplicons=98, seed=7):
    """Returns (primers, amplicons): primers = sorted [(start, end, name)], amplicons =
    int array [n,4] of (amp_start, amp_end, left_primer_end, right_primer_start)."""
    rng = np.random.default_rng(seed)
    amp_len = rng.integers(360, 391, size=n_amplicons)
    first, last_end = 30, length - 30
    starts = np.round(np.linspace(first, last_end - amp_len[-1], n_amplicons)).astype(np.int64)
    amps = np.zeros((n_amplicons, 4), dtype=np.int64)
    primers = []
    for i in range(n_amplicons):
        a0 = int(starts[i]); a1 = int(min(a0 + amp_len[i], length - 1))
        ll = int(rng.integers(22, 31)); lr = int(rng.integers(22, 31))
        amps[i] = (a0, a1, a0 + ll, a1 - lr)
        primers.append((a0, a0 + ll, "SYN_%d_LEFT" % (i + 1)))
        primers.append((a1 - lr, a1, "SYN_%d_RIGHT" % (i + 1)))
    primers.sort()
    return primers, amps


def write_bed(path, primers, chrom="SYN_REF"):
    with open(path, "w") as f:
        for s, e, name in primers:
            f.write("%s\t%d\t%d\t%s\n" % (chrom, s, e, name))


def _draw_quals(rng, n, L):
    idx = np.searchsorted(np.cumsum(QUAL_PROBS), rng.random((n, L)), side="right")
    return QUAL_VALUES[np.minimum(idx, 3)]


def make_amplicon_batch(genome, amps, n_reads, seed, read_len=150, indel_frac=0.10,
                        sub_rate=0.005, lowq_tail_frac=0.15, chunk=262144):
    """Config 2/3 generator: returns a coordinate-sorted ReadBatch of ``n_reads`` reads."""
    rng = np.random.default_rng(seed)
    G = genome.size
    L = read_len
    parts = []
    done = 0
    while done < n_reads:
        n = min(chunk, n_reads - done)
        done += n
        a = rng.integers(0, amps.shape[0], size=n)
        rev = rng.random(n) < 0.5
        jit = rng.integers(-3, 4, size=n)
        kind = np.zeros(n, np.int8)  # 0 none, 1 insertion, 2 deletion
        has = rng.random(n) < indel_frac
        kind[has] = rng.integers(1, 3, size=int(has.sum()))
        k = rng.integers(1, 4, size=n)
        k[kind == 0] = 0
        off = rng.integers(5, L - 5 - 3, size=n)  # interior offset, >=5 from either end
        ref_span = L + np.where(kind == 2, k, 0) - np.where(kind == 1, k, 0)
        pos = np.where(rev, amps[a, 1] + jit - ref_span, amps[a, 0] + jit)
        pos = np.clip(pos, 0, G - ref_span - 1)
        j = np.arange(L)[None, :]
        # reference index of every query base (insertion bases get a dummy index)
        shift = np.where(kind[:, None] == 2, np.where(j >= off[:, None], k[:, None], 0), 0)
        shift = shift - np.where(kind[:, None] == 1,
                                 np.clip(j - off[:, None], 0, k[:, None]), 0)
        ridx = pos[:, None] + j + shift
        codes = genome[np.clip(ridx, 0, G - 1)]
        ins_mask = (kind[:, None] == 1) & (j >= off[:, None]) & (j < (off + k)[:, None])
        codes = np.where(ins_mask, ACGT_CODES[rng.integers(0, 4, size=(n, L))], codes)
        sub = rng.random((n, L)) < sub_rate
        if sub.any():
            cur = np.searchsorted(ACGT_CODES, codes[sub])
            codes[sub] = ACGT_CODES[(cur + rng.integers(1, 4, size=cur.size)) % 4]
        qual = _draw_quals(rng, n, L)
        tail = np.where(rng.random(n) < lowq_tail_frac, rng.integers(0, 21, size=n), 0)
        low = np.where(rev[:, None], j < tail[:, None], j >= (L - tail)[:, None])
        qual = np.where(low, np.uint8(2), qual)
        flag = np.where(rev, np.where(rng.random(n) < 0.5, 147, 83),
                        np.where(rng.random(n) < 0.5, 99, 163)).astype(np.uint16)
        alen = (amps[a, 1] - amps[a, 0]).astype(np.int64)
        tlen = np.where(rev, -alen, alen).astype(np.int32)
        ops = np.zeros((n, 3), np.uint32)
        nops = np.where(kind == 0, 1, 3).astype(np.int64)
        ops[:, 0] = np.where(kind == 0, (L << 4) | OP_M, (off << 4) | OP_M)
        ops[:, 1] = (k << 4) | np.where(kind == 1, OP_I, OP_D)
        ops[:, 2] = ((L - off - np.where(kind == 1, k, 0)) << 4) | OP_M
        parts.append((pos.astype(np.int32), flag, tlen, ops, nops, codes.astype(np.uint8),
                      qual.astype(np.uint8)))
    pos = np.concatenate([p[0] for p in parts]); flag = np.concatenate([p[1] for p in parts])
    tlen = np.concatenate([p[2] for p in parts]); ops = np.concatenate([p[3] for p in parts])
    nops = np.concatenate([p[4] for p in parts]); codes = np.concatenate([p[5] for p in parts])
    qual = np.concatenate([p[6] for p in parts])
    order = np.argsort(pos, kind="stable")
    pos, flag, tlen, ops, nops = pos[order], flag[order], tlen[order], ops[order], nops[order]
    codes, qual = codes[order], qual[order]
    cig_off = np.zeros(n_reads + 1, np.uint64)
    cig_off[1:] = np.cumsum(nops)
    cig = ops[np.arange(3)[None, :] < nops[:, None]]
    return ReadBatch.from_uniform(pos, flag, tlen, L, cig_off, cig, codes, qual)


def reads_for_depth(depth, read_len=150, genome_len=GENOME_LEN):
    """Read count giving ``depth`` x mean coverage (1k x -> 199,353 for 150 bp)."""
    return int(round(depth * genome_len / read_len))


# ---------------------------------------------------------------------------
# Ragged generators (per-read Python loops; used for config 5 pools and fixtures)
# ---------------------------------------------------------------------------

def _rand_seq(rng, n, n_rate=0.0, iupac_rate=0.0):
    s = np.array(list("ACGT"))[rng.integers(0, 4, size=n)]
    if n_rate:
        s[rng.random(n) < n_rate] = "N"
    if iupac_rate:
        s[rng.random(n) < iupac_rate] = "R"
    return "".join(s)


def make_mixed_segments(genome, amps, n_reads, seed):
    """Config 5 pool: 75-300 bp reads, 40 % soft-clipped, 40 % with 2-6 indel ops (1-12 bp),
    I never first/last nor adjacent to D.  Returns coordinate-sorted Segments."""
    rng = np.random.default_rng(seed)
    G = genome.size
    gstr = genome_string(genome)
    segs = []
    for _ in range(n_reads):
        L = int(rng.integers(75, 301))
        lead = int(rng.integers(10, 61)) if rng.random() < 0.28 else 0
        trail = int(rng.integers(10, 61)) if rng.random() < 0.28 else 0
        if lead + trail > L - 30:
            lead = trail = 0
        body = L - lead - trail
        ops = []
        if rng.random() < 0.40:
            n_ind = int(rng.integers(2, 7))
            kinds = []
            for _k in range(n_ind):
                kinds.append(OP_I if rng.random() < 0.5 else OP_D)
            lens = rng.integers(1, 13, size=n_ind)
            q_ins = int(sum(l for kd, l in zip(kinds, lens) if kd == OP_I))
            m_total = body - q_ins
            if m_total < 4 * (n_ind + 1):
                kinds = []; m_total = body
            if kinds:
                cuts = np.sort(rng.choice(np.arange(1, m_total // 4), size=n_ind, replace=False)) * 4
                m_lens = np.diff(np.concatenate([[0], cuts, [m_total]]))
                for i, kd in enumerate(kinds):
                    ops.append((OP_M, int(m_lens[i]))); ops.append((kd, int(lens[i])))
                ops.append((OP_M, int(m_lens[-1])))
            else:
                ops.append((OP_M, body))
        else:
            ops.append((OP_M, body))
        ref_span = sum(n for o, n in ops if o in (OP_M, OP_D))
        a = int(rng.integers(0, amps.shape[0]))
        rev = rng.random() < 0.5
        jit = int(rng.integers(-3, 4))
        pos = (amps[a, 1] + jit - ref_span) if rev else (amps[a, 0] + jit)
        pos = int(min(max(pos, 0), G - ref_span - 1))
        seq = []
        r = pos
        for o, n in ops:
            if o == OP_M:
                seq.append(gstr[r:r + n]); r += n
            elif o == OP_I:
                seq.append(_rand_seq(rng, n))
            else:
                r += n
        seq = _rand_seq(rng, lead) + "".join(seq) + _rand_seq(rng, trail)
        seq = list(seq)
        for i in np.nonzero(rng.random(L) < 0.005)[0]:
            seq[i] = "ACGT"[("ACGT".index(seq[i]) + int(rng.integers(1, 4))) % 4]
        qual = _draw_quals(rng, 1, L)[0]
        if rng.random() < 0.15:
            t = int(rng.integers(0, 21))
            if t:
                if rev:
                    qual[:t] = 2
                else:
                    qual[L - t:] = 2
        cig = ([(OP_S, lead)] if lead else []) + ops + ([(OP_S, trail)] if trail else [])
        flag = int(rng.choice([147, 83]) if rev else rng.choice([99, 163]))
        alen = int(amps[a, 1] - amps[a, 0])
        segs.append(Segment(flag=flag, reference_start=pos, cigar=cig,
                            template_length=-alen if rev else alen,
                            query_sequence="".join(seq), query_qualities=qual.tolist()))
    segs.sort(key=lambda s: s.reference_start)
    return segs


def random_segments(rng, n, ref_len, primers, weird=0.15, domain_errors=True, max_len=160):
    """Adversarial reads for parity tests: every CIGAR op, clips, indels at the edges,
    low-quality runs, reads inside primers, and (optionally) inputs on which the reference
    raises."""
    segs = []
    primers = list(primers)
    for _ in range(n):
        L = int(rng.integers(12, max_len + 1))
        is_weird = rng.random() < weird
        ops = []
        q_left = L
        if rng.random() < 0.10:
            ops.append((OP_H, int(rng.integers(1, 40))))
        if rng.random() < 0.30 and q_left > 20:
            k = int(rng.integers(1, min(40, q_left - 10))); ops.append((OP_S, k)); q_left -= k
        trail_s = 0
        if rng.random() < 0.30 and q_left > 20:
            trail_s = int(rng.integers(1, min(40, q_left - 10))); q_left -= trail_s
        body = []
        n_body = int(rng.integers(1, 7)) if rng.random() < 0.5 else 1
        remaining = q_left
        for b in range(n_body):
            last = b == n_body - 1
            if is_weird:
                op = int(rng.choice([OP_M, OP_I, OP_D, OP_N, OP_EQ, OP_X, OP_P, OP_M, OP_M]))
            elif b % 2 == 0:
                op = int(rng.choice([OP_M, OP_M, OP_M, OP_EQ, OP_X]))
            else:
                op = int(rng.choice([OP_I, OP_D, OP_D, OP_N, OP_I]))
            if last and not is_weird and op in (OP_I, OP_D, OP_N):
                op = OP_M
            if op in (OP_D, OP_N):
                body.append((op, int(rng.integers(1, 13)))); continue
            if op == OP_P:
                body.append((op, int(rng.integers(1, 4)))); continue
            if remaining <= 0:
                break
            if last:
                k = remaining
            elif op == OP_I:
                k = int(rng.integers(1, min(12, remaining) + 1))
            else:
                k = int(rng.integers(1, remaining + 1))
            body.append((op, k)); remaining -= k
        if remaining > 0:
            body.append((OP_M, remaining))
        ops.extend(body)
        if trail_s:
            ops.append((OP_S, trail_s))
        if rng.random() < 0.10:
            ops.append((OP_H, int(rng.integers(1, 40))))
        ref_span = sum(k for o, k in ops if o in (OP_M, OP_D, OP_N, OP_EQ, OP_X))
        # place the read next to a primer so trimming is exercised
        ps, pe = primers[int(rng.integers(0, len(primers)))][:2]
        mode = rng.random()
        if mode < 0.40:
            pos = ps + int(rng.integers(-8, pe - ps + 8))
        elif mode < 0.80:
            pos = pe + int(rng.integers(-8, 9)) - max(ref_span, 1)
        else:
            pos = int(rng.integers(0, ref_len))
        hi = ref_len - max(ref_span, 1)
        if not (domain_errors and rng.random() < 0.02):
            pos = min(pos, hi)
        pos = max(pos, 0)
        seq = _rand_seq(rng, L, n_rate=0.01, iupac_rate=0.0004 if domain_errors else 0.0)
        qmode = rng.random()
        qual = rng.integers(21, 41, size=L)
        if qmode < 0.45:
            for _r in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(1, 25)))
                qual[a:b] = rng.integers(0, 20, size=b - a)
        elif qmode < 0.55:
            t = int(rng.integers(1, min(L, 40)))
            if rng.random() < 0.5:
                qual[:t] = rng.integers(0, 15, size=t)
            else:
                qual[L - t:] = rng.integers(0, 15, size=t)
        elif qmode < 0.58:
            qual[:] = rng.integers(0, 19, size=L)
        elif qmode < 0.70:
            qual = rng.integers(0, 42, size=L)
        flag = int(rng.choice([0, 16, 99, 147, 83, 163, 1, 17, 2145, 2064]))
        big = rng.random() < 0.5
        tl = int(rng.integers(200, 500)) if big else int(rng.integers(0, 170))
        if rng.random() < 0.5:
            tl = -tl
        segs.append(Segment(flag=flag, reference_start=pos, cigar=ops, template_length=tl,
                            query_sequence=seq, query_qualities=qual.tolist()))
    return segs


def gather_rows(b, idx):
    """The rows ``idx`` of batch ``b`` as a new batch (variable-length CIGARs / bases gathered with repeat + cumsum)."""
    def spans(off):
        ln = (off[1:] - off[:-1]).astype(np.int64)[idx]
        new_off = np.zeros(idx.size + 1, np.int64); np.cumsum(ln, out=new_off[1:])
        src = np.repeat(off[:-1].astype(np.int64)[idx] - new_off[:-1], ln) + np.arange(int(new_off[-1]), dtype=np.int64)
        return new_off, src
    co, csrc = spans(b.cig_off)
    so, ssrc = spans(b.seq_off)
    seq_nib = np.empty(b.seq.size * 2, np.uint8); seq_nib[0::2] = b.seq >> 4; seq_nib[1::2] = b.seq & 15
    nib = seq_nib[ssrc]
    return ReadBatch(b.pos[idx], b.flag[idx], b.tlen[idx], b.lseq[idx], co.astype(np.uint64), b.cig[csrc], so.astype(np.uint64),
                     ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8), b.qual[ssrc])


def make_config5_batch(genome, amps, rep=200, pool_reads=40000, seed=3):
    """BASELINE config 5 at size: mixed 75-300 bp reads with long soft clips and indel-heavy CIGARs, ``rep`` x ``pool_reads``
    reads (200 x 40,000 = 8.0 M = 50k x depth).  The pool of make_mixed_segments is gathered ``rep`` times with numpy, every
    copy shifted by 0..7 positions (seeded) and the whole sorted again, so copies of different reads interleave like the
    reads of a real pile (``rep`` identical reads in a row would make every tile homogeneous) and packing takes seconds."""
    pool = ReadBatch.from_segments(sorted(make_mixed_segments(genome, amps, pool_reads, seed=seed), key=lambda s: s.reference_start))
    rng = np.random.default_rng(5)
    idx = np.repeat(np.arange(pool.n, dtype=np.int64), rep)
    jit = rng.integers(0, 8, idx.size).astype(np.int32)
    order = np.argsort(pool.pos[idx].astype(np.int64) + jit, kind="stable")
    idx, jit = idx[order], jit[order]
    b = gather_rows(pool, idx)
    w = pool.cig
    refop = np.isin(w & 15, (0, 2, 3, 7, 8))
    cum = np.concatenate([[0], np.cumsum((w >> 4).astype(np.int64) * refop)])
    span = cum[pool.cig_off[1:].astype(np.int64)] - cum[pool.cig_off[:-1].astype(np.int64)]
    ok = pool.pos[idx].astype(np.int64) + jit + span[idx] < genome.size      # (a shifted copy must still end inside the reference)
    b.pos[:] = pool.pos[idx] + np.where(ok, jit, 0).astype(np.int32)
    return b
