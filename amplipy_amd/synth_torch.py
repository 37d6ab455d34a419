"""Synthetic amplicon batches generated directly in HBM with torch (bench plumbing).

Same distribution as ``synth.make_amplicon_batch`` but
built on the GPU so that 20 M-read batches take seconds, and laid out in the device format of
``amp_dev_reads`` (32-bit offsets).  torch is used for device memory and RNG only.
"""
from __future__ import annotations

import numpy as np
import torch

from . import abi
from .batch import ReadBatch

QUAL_VALUES = (37, 25, 11, 2)
QUAL_CUM = (0.80, 0.92, 0.98)


class DeviceBatch:
    """Device-resident read batch (torch tensors) + the AmpDevReads struct pointing at them."""

    def __init__(self, n, pos, flag, tlen, lseq, cig_off32, cig, seq_off8, seq, qual, n_cig, n_bases_padded):
        self.n = int(n)
        self.pos, self.flag, self.tlen, self.lseq = pos, flag, tlen, lseq
        self.cig_off32, self.cig, self.seq_off8, self.seq, self.qual = cig_off32, cig, seq_off8, seq, qual
        self.n_cig, self.n_bases_padded = int(n_cig), int(n_bases_padded)

    def struct(self):
        return abi.AmpDevReads(self.n, self.pos.data_ptr(), self.flag.data_ptr(), self.tlen.data_ptr(),
                               self.lseq.data_ptr(), self.cig_off32.data_ptr(), self.cig.data_ptr(),
                               self.seq_off8.data_ptr(), self.seq.data_ptr(), self.qual.data_ptr(),
                               self.n_cig, self.n_bases_padded)

    def rows(self, lo, hi):
        """Rows [lo, hi) as a batch that shares this one's CIGAR / base / quality arrays (offsets stay absolute:
        the kernels index those arrays through cig_off32 / seq_off8).  n_cig is the END offset of the last row, so
        that an output CIGAR array of n_cig + 3 n words has room for the slots cig_off32[i] + 3 i; n_bases_padded is the
        rows' own (the library picks the fast kernel's tile geometry by the mean padded row length)."""
        return DeviceBatch(hi - lo, self.pos[lo:hi], self.flag[lo:hi], self.tlen[lo:hi], self.lseq[lo:hi],
                           self.cig_off32[lo:hi + 1], self.cig, self.seq_off8[lo:hi + 1], self.seq, self.qual,
                           int(self.cig_off32[hi].item()), int((self.seq_off8[hi] - self.seq_off8[lo]).item()) * 8)

    def to_host(self, lo=0, hi=None):
        """Rows [lo, hi) as a host ReadBatch (for the CPU baseline / parity check)."""
        hi = self.n if hi is None else hi
        co = self.cig_off32[lo:hi + 1].cpu().numpy().astype(np.int64).view(np.int64)
        so = self.seq_off8[lo:hi + 1].cpu().numpy().astype(np.int64) * 8
        c0, c1, s0, s1 = int(co[0]), int(co[-1]), int(so[0]), int(so[-1])
        return ReadBatch(self.pos[lo:hi].cpu().numpy(), self.flag[lo:hi].cpu().numpy().view(np.uint16),
                         self.tlen[lo:hi].cpu().numpy(), self.lseq[lo:hi].cpu().numpy().view(np.uint32),
                         (co - c0).astype(np.uint64), self.cig[c0:c1].cpu().numpy().view(np.uint32),
                         (so - s0).astype(np.uint64), self.seq[s0 // 2:s1 // 2].cpu().numpy(),
                         self.qual[s0:s1].cpu().numpy())

    @classmethod
    def from_host(cls, batch, device):
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(device)
        pad = lambda x: torch.cat([x, torch.zeros(16, dtype=x.dtype, device=device)])
        return cls(batch.n, t(batch.pos, np.int32), t(batch.flag, np.int16), t(batch.tlen, np.int32),
                   t(batch.lseq, np.int32), t(batch.cig_off.astype(np.uint32), np.int32), t(batch.cig, np.int32),
                   t((batch.seq_off // 8).astype(np.uint32), np.int32), pad(t(batch.seq, np.uint8)),
                   pad(t(batch.qual, np.uint8)), batch.cig.size, int(batch.seq_off[-1]))


def make_amplicon_batch_device(genome, amps, n_reads, seed, device, read_len=150, indel_frac=0.10,
                               sub_rate=0.005, lowq_tail_frac=0.15, chunk=1 << 20):
    """Coordinate-sorted synthetic batch of ``n_reads`` reads drawn from the amplicons ``amps``."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    G = int(genome.size)
    L = int(read_len)
    stride = L + ((-L) % 8)
    dev = torch.device(device)
    gen = torch.from_numpy(np.ascontiguousarray(genome)).to(dev)
    amps_t = torch.from_numpy(np.ascontiguousarray(amps[:, :2]).astype(np.int64)).to(dev)
    n = int(n_reads)
    rnd = lambda *s: torch.rand(*s, generator=g, device=dev)
    rint = lambda lo, hi, *s: torch.randint(lo, hi, s, generator=g, device=dev)

    a = rint(0, amps_t.shape[0], n)
    rev = rnd(n) < 0.5
    jit = rint(-3, 4, n)
    has = rnd(n) < indel_frac
    kind = torch.where(has, rint(1, 3, n), torch.zeros(n, dtype=torch.int64, device=dev))
    k = torch.where(kind > 0, rint(1, 4, n), torch.zeros(n, dtype=torch.int64, device=dev))
    off = rint(5, L - 5 - 3, n)
    ref_span = L + torch.where(kind == 2, k, 0) - torch.where(kind == 1, k, 0)
    pos = torch.where(rev, amps_t[a, 1] + jit - ref_span, amps_t[a, 0] + jit)
    pos = torch.minimum(torch.clamp(pos, min=0), G - ref_span - 1)
    alen = amps_t[a, 1] - amps_t[a, 0]
    tlen = torch.where(rev, -alen, alen)
    flag = torch.where(rev, torch.where(rnd(n) < 0.5, 147, 83), torch.where(rnd(n) < 0.5, 99, 163))
    tail = torch.where(rnd(n) < lowq_tail_frac, rint(0, 21, n), torch.zeros(n, dtype=torch.int64, device=dev))
    order = torch.sort(pos, stable=True).indices
    pos, rev, kind, k, off, tlen, flag, tail = (x[order] for x in (pos, rev, kind, k, off, tlen, flag, tail))

    nops = torch.where(kind == 0, 1, 3)
    cig_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    cig_off[1:] = torch.cumsum(nops, 0)
    ops = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    ops[:, 0] = torch.where(kind == 0, (L << 4) | 0, (off << 4) | 0)
    ops[:, 1] = (k << 4) | torch.where(kind == 1, 1, 2)
    ops[:, 2] = ((L - off - torch.where(kind == 1, k, 0)) << 4) | 0
    cig = ops[torch.arange(3, device=dev)[None, :] < nops[:, None]].to(torch.int32)

    seq = torch.zeros(n * stride // 2 + 16, dtype=torch.uint8, device=dev)   # +16: slack for vector loads
    qual = torch.zeros(n * stride + 16, dtype=torch.uint8, device=dev)
    acgt = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)
    qv = torch.tensor(QUAL_VALUES, dtype=torch.uint8, device=dev)
    qcum = torch.tensor(QUAL_CUM, device=dev)
    j = torch.arange(L, device=dev)[None, :]
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        kd, kk, of, ps, rv, tl = (x[lo:hi, None] for x in (kind, k, off, pos, rev, tail))
        shift = torch.where(kd == 2, torch.where(j >= of, kk, 0), 0) - \
            torch.where(kd == 1, torch.clamp(j - of, min=0).minimum(kk), 0)
        codes = gen[torch.clamp(ps + j + shift, 0, G - 1)]
        ins_mask = (kd == 1) & (j >= of) & (j < of + kk)
        codes = torch.where(ins_mask, acgt[rint(0, 4, m, L)], codes)
        sub = rnd(m, L) < sub_rate
        cur = (codes.to(torch.int64) > 1).to(torch.int64) + (codes.to(torch.int64) > 2) + (codes.to(torch.int64) > 4)
        codes = torch.where(sub, acgt[(cur + rint(1, 4, m, L)) % 4], codes)
        q = qv[torch.bucketize(rnd(m, L), qcum, right=True)]
        low = torch.where(rv, j < tl, j >= (L - tl))
        q = torch.where(low, torch.tensor(2, dtype=torch.uint8, device=dev), q)
        cpad = torch.zeros((m, stride), dtype=torch.uint8, device=dev)
        cpad[:, :L] = codes
        seq[lo * stride // 2:hi * stride // 2] = ((cpad[:, 0::2] << 4) | cpad[:, 1::2]).reshape(-1)
        qpad = torch.zeros((m, stride), dtype=torch.uint8, device=dev)
        qpad[:, :L] = q
        qual[lo * stride:hi * stride] = qpad.reshape(-1)
    seq_off8 = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (stride // 8)).to(torch.int32)
    return DeviceBatch(n, pos.to(torch.int32), flag.to(torch.int16), tlen.to(torch.int32),
                       torch.full((n,), L, dtype=torch.int32, device=dev), cig_off.to(torch.int32), cig, seq_off8,
                       seq, qual, int(cig_off[-1].item()), n * stride)
