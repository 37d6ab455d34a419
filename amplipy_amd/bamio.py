"""SAM / BAM input and output for the host mirror.

AmpliPy leaves file I/O to pysam (AmpliPy.py:296-360, :896, :911); pysam / htslib are not
available to this build, so the container formats are implemented here from the SAM/BAM
specification: BGZF (gzip members with a BC extra field, zlib raw deflate), BAM records
(32-byte core + name + uint32 CIGAR + 4-bit SEQ + QUAL + aux) and SAM text.

A record is kept as a ``Rec``: the SAM-level fields the hot path needs plus the untouched
remainder (name, mate fields, aux) so that a trimmed record can be written back with only POS,
CIGAR (and BAM bin) changed -- what ``out_aln.write(s)`` does after trim_read edited ``s``.
"""
from __future__ import annotations

import struct
import sys
import zlib

import numpy as np

from .batch import SEQ_NT16
from .segment import CIGAR_OPS, format_cigar, parse_cigar

BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_NT16_ENC = {c: i for i, c in enumerate(SEQ_NT16)}
_NT16_ENC.update({c.lower(): i for c, i in list(_NT16_ENC.items())})


class Rec:
    __slots__ = ("qname", "flag", "ref_id", "pos", "mapq", "cigar", "next_ref_id", "next_pos", "tlen", "seq",
                 "qual", "aux_bam", "aux_sam")

    def __init__(self, qname, flag, ref_id, pos, mapq, cigar, next_ref_id, next_pos, tlen, seq, qual,
                 aux_bam=None, aux_sam=None):
        self.qname = qname; self.flag = flag; self.ref_id = ref_id; self.pos = pos; self.mapq = mapq
        self.cigar = cigar          # list[(op, len)] or None
        self.next_ref_id = next_ref_id; self.next_pos = next_pos; self.tlen = tlen
        self.seq = seq              # str or None ('*')
        self.qual = qual            # bytes of Phred values or None ('*')
        self.aux_bam = aux_bam      # raw BAM aux bytes (when read from BAM)
        self.aux_sam = aux_sam      # list of 'TAG:T:V' strings (when read from SAM)

    def to_segment(self):
        from .segment import Segment
        return Segment(flag=self.flag, reference_start=self.pos, cigar=self.cigar, template_length=self.tlen,
                       query_sequence=self.seq, query_qualities=None if self.qual is None else list(self.qual),
                       query_name=self.qname)


class Header:
    def __init__(self, text, refs):
        self.text = text            # SAM header text ('@..' lines, newline-terminated)
        self.refs = refs            # [(name, length)]

    def to_dict_pg(self):
        """The @PG entries as dicts, in order (header_dict['PG'] of AmpliPy.py:330)."""
        out = []
        for line in self.text.splitlines():
            if line.startswith("@PG"):
                out.append({f[:2]: f[3:] for f in line.split("\t")[1:] if len(f) >= 3 and f[2] == ":"})
        return out

    def with_amplipy_pg(self, version, command_line):
        """Header after AmpliPy.py:330-342: append an @PG whose ID is AmpliPy / AmpliPy.<k> and whose
        PP is the last existing @PG's ID.  Raises KeyError like the reference when there is no @PG."""
        pgs = self.to_dict_pg()
        if not pgs:
            raise KeyError("PG")
        n_prev = sum(1 for p in pgs if p.get("PN") == "AmpliPy")
        pg_id = "AmpliPy" if n_prev == 0 else "AmpliPy.%d" % n_prev
        line = "@PG\tID:%s\tPN:AmpliPy\tPP:%s\tVN:%s\tCL:%s\n" % (pg_id, pgs[-1]["ID"], version, command_line)
        text = self.text if self.text.endswith("\n") or not self.text else self.text + "\n"
        return Header(text + line, self.refs)


def _refs_from_text(text):
    refs = []
    for line in text.splitlines():
        if line.startswith("@SQ"):
            d = {f[:2]: f[3:] for f in line.split("\t")[1:]}
            refs.append((d.get("SN", "*"), int(d.get("LN", "0"))))
    return refs


# ---------------------------------------------------------------------------------------------
# aux conversion
# ---------------------------------------------------------------------------------------------
_B_TYPES = {"c": ("b", 1), "C": ("B", 1), "s": ("h", 2), "S": ("H", 2), "i": ("i", 4), "I": ("I", 4), "f": ("f", 4)}


def aux_bam_to_sam(buf):
    out = []
    i = 0
    n = len(buf)
    while i + 3 <= n:
        tag = buf[i:i + 2].decode("ascii"); t = chr(buf[i + 2]); i += 3
        if t == "A":
            out.append("%s:A:%s" % (tag, chr(buf[i]))); i += 1
        elif t in _B_TYPES:
            fmt, sz = _B_TYPES[t]
            v = struct.unpack_from("<" + fmt, buf, i)[0]; i += sz
            out.append("%s:%s:%s" % (tag, "f" if t == "f" else "i", ("%g" % v) if t == "f" else v))
        elif t in "ZH":
            j = buf.index(0, i)
            out.append("%s:%s:%s" % (tag, t, buf[i:j].decode("ascii"))); i = j + 1
        elif t == "B":
            st = chr(buf[i]); cnt = struct.unpack_from("<I", buf, i + 1)[0]; i += 5
            fmt, sz = _B_TYPES[st]
            vals = struct.unpack_from("<%d%s" % (cnt, fmt), buf, i); i += cnt * sz
            out.append("%s:B:%s%s" % (tag, st, "".join("," + (("%g" % v) if st == "f" else str(v)) for v in vals)))
        else:
            raise ValueError("unknown BAM aux type %r" % t)
    return out


def aux_sam_to_bam(fields):
    out = bytearray()
    for f in fields:
        tag, t, v = f[:2], f[3], f[5:]
        out += tag.encode("ascii")
        if t == "A":
            out += b"A" + v.encode("ascii")
        elif t == "i":
            x = int(v)
            for code, lo, hi in (("c", -128, 127), ("C", 0, 255), ("s", -32768, 32767), ("S", 0, 65535),
                                 ("i", -2 ** 31, 2 ** 31 - 1), ("I", 0, 2 ** 32 - 1)):
                if lo <= x <= hi:
                    out += code.encode() + struct.pack("<" + _B_TYPES[code][0], x); break
        elif t == "f":
            out += b"f" + struct.pack("<f", float(v))
        elif t in "ZH":
            out += t.encode() + v.encode("ascii") + b"\0"
        elif t == "B":
            st = v[0]; vals = [x for x in v[2:].split(",") if x != ""] if len(v) > 1 else []
            fmt = _B_TYPES[st][0]
            conv = float if st == "f" else int
            out += b"B" + st.encode() + struct.pack("<I", len(vals)) + struct.pack("<%d%s" % (len(vals), fmt), *[conv(x) for x in vals])
        else:
            raise ValueError("unknown SAM aux type %r" % t)
    return bytes(out)


# ---------------------------------------------------------------------------------------------
# BGZF
# ---------------------------------------------------------------------------------------------
def bgzf_blocks(f):
    """Yield the decompressed payload of every BGZF block of an open binary file."""
    while True:
        hdr = f.read(18)
        if len(hdr) == 0:
            return
        if len(hdr) < 18 or hdr[:4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF stream")
        xlen = struct.unpack_from("<H", hdr, 10)[0]
        extra = hdr[12:18] + f.read(xlen - 6)
        bsize = None
        i = 0
        while i + 4 <= len(extra):
            si1, si2, slen = extra[i], extra[i + 1], struct.unpack_from("<H", extra, i + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", extra, i + 4)[0]
            i += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        rest = f.read(bsize + 1 - 12 - xlen)
        data = zlib.decompress(rest[:-8], -15)
        yield data


class BgzfWriter:
    def __init__(self, f, level=6):
        self.f = f; self.buf = bytearray(); self.level = level

    def write(self, b):
        self.buf += b
        while len(self.buf) >= 0xFF00:
            self._flush_block(bytes(self.buf[:0xFF00])); del self.buf[:0xFF00]

    def _flush_block(self, data):
        c = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        bsize = len(comp) + 25
        self.f.write(struct.pack("<4BIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize))
        self.f.write(comp)
        self.f.write(struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    def close(self):
        if self.buf:
            self._flush_block(bytes(self.buf)); self.buf = bytearray()
        self.f.write(BGZF_EOF)
        self.f.close()


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


# ---------------------------------------------------------------------------------------------
# readers
# ---------------------------------------------------------------------------------------------
class AlignmentReader:
    """Iterates Rec objects of a SAM (text) or BAM file, in file order, mapped or not."""

    def __init__(self, path, mode):
        self.path = path; self.mode = mode
        if mode == "rb":
            self._f = open(path, "rb")
            self._blocks = bgzf_blocks(self._f)
            self._buf = bytearray()
            magic = self._take(4)
            if magic != b"BAM\1":
                raise ValueError("not a BAM file: %s" % path)
            l_text = struct.unpack("<i", self._take(4))[0]
            text = self._take(l_text).rstrip(b"\0").decode("utf-8")
            n_ref = struct.unpack("<i", self._take(4))[0]
            refs = []
            for _ in range(n_ref):
                l_name = struct.unpack("<i", self._take(4))[0]
                name = self._take(l_name)[:-1].decode("ascii")
                refs.append((name, struct.unpack("<i", self._take(4))[0]))
            self.header = Header(text, refs)
        else:
            self._f = sys.stdin if path == "-" else open(path, "r")
            lines = []
            self._first = None
            for line in self._f:
                if line.startswith("@"):
                    lines.append(line if line.endswith("\n") else line + "\n")
                else:
                    self._first = line; break
            text = "".join(lines)
            self.header = Header(text, _refs_from_text(text))
        self._ref_index = {n: i for i, (n, _) in enumerate(self.header.refs)}

    def _take(self, n):
        while len(self._buf) < n:
            try:
                self._buf += next(self._blocks)
            except StopIteration:
                break
        out = bytes(self._buf[:n]); del self._buf[:n]
        return out

    def __iter__(self):
        return self._iter_bam() if self.mode == "rb" else self._iter_sam()

    def _iter_bam(self):
        while True:
            h = self._take(4)
            if len(h) < 4:
                return
            bs = struct.unpack("<i", h)[0]
            b = self._take(bs)
            ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
            o = 32
            qname = b[o:o + l_name - 1].decode("ascii"); o += l_name
            cig = None
            if n_cig:
                raw = struct.unpack_from("<%dI" % n_cig, b, o)
                cig = [(v & 15, v >> 4) for v in raw]
            o += 4 * n_cig
            seq = None
            if l_seq:
                packed = np.frombuffer(b, np.uint8, (l_seq + 1) // 2, o)
                codes = np.empty(packed.size * 2, np.uint8); codes[0::2] = packed >> 4; codes[1::2] = packed & 15
                seq = "".join(SEQ_NT16[c] for c in codes[:l_seq])
            o += (l_seq + 1) // 2
            qual = b[o:o + l_seq] if l_seq else None
            if qual is not None and l_seq and qual[0] == 0xFF:
                qual = None
            o += l_seq
            yield Rec(qname, flag, ref_id, pos, mapq, cig, nref, npos, tlen, seq, qual, aux_bam=b[o:])

    def _iter_sam(self):
        import itertools
        src = itertools.chain([self._first], self._f) if self._first is not None else iter(())
        for line in src:
            f = line.rstrip("\r\n").split("\t")
            if len(f) < 11:
                continue
            rname = f[2]; rnext = f[6]
            ref_id = self._ref_index.get(rname, -1) if rname != "*" else -1
            nref = ref_id if rnext == "=" else (self._ref_index.get(rnext, -1) if rnext != "*" else -1)
            seq = None if f[9] == "*" else f[9]
            qual = None if f[10] == "*" else bytes(ord(c) - 33 for c in f[10])
            yield Rec(f[0], int(f[1]), ref_id, int(f[3]) - 1, int(f[4]), parse_cigar(f[5]), nref, int(f[7]) - 1,
                      int(f[8]), seq, qual, aux_sam=f[11:])

    def close(self):
        if self._f is not sys.stdin:
            self._f.close()


# ---------------------------------------------------------------------------------------------
# writers
# ---------------------------------------------------------------------------------------------
class AlignmentWriter:
    def __init__(self, path, mode, header):
        self.mode = mode; self.header = header
        if mode == "wb":
            self._w = BgzfWriter(open(path, "wb"))
            text = header.text.encode("utf-8")
            out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(header.refs)))
            for name, ln in header.refs:
                nb = name.encode("ascii") + b"\0"
                out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
            self._w.write(bytes(out))
        else:
            self._f = sys.stdout if path == "-" else open(path, "w")
            self._f.write(header.text)

    def _ref_name(self, i):
        return "*" if i < 0 else self.header.refs[i][0]

    def write(self, r, pos=None, cigar=None):
        """Write record ``r`` with (optionally) a new 0-based POS and CIGAR."""
        pos = r.pos if pos is None else pos
        cigar = r.cigar if cigar is None else cigar
        if self.mode == "wb":
            name = r.qname.encode("ascii") + b"\0"
            l_seq = 0 if r.seq is None else len(r.seq)
            rlen = sum(n for op, n in (cigar or ()) if op in (0, 2, 3, 7, 8))
            end = pos + (rlen if rlen else 1)
            core = struct.pack("<iiBBHHHiiii", r.ref_id, pos, len(name), r.mapq, reg2bin(max(pos, 0), max(end, 1)),
                               len(cigar or ()), r.flag, l_seq, r.next_ref_id, r.next_pos, r.tlen)
            cg = struct.pack("<%dI" % len(cigar or ()), *[(n << 4) | op for op, n in (cigar or ())])
            sq = b""
            if l_seq:
                codes = [_NT16_ENC.get(c, 15) for c in r.seq]
                if l_seq & 1:
                    codes.append(0)
                sq = bytes((codes[k] << 4) | codes[k + 1] for k in range(0, len(codes), 2))
            ql = (bytes(r.qual) if r.qual is not None else b"\xff" * l_seq)
            aux = r.aux_bam if r.aux_bam is not None else aux_sam_to_bam(r.aux_sam or [])
            body = core + name + cg + sq + ql + aux
            self._w.write(struct.pack("<i", len(body)) + body)
        else:
            rn = self._ref_name(r.ref_id)
            rnext = "*" if r.next_ref_id < 0 else ("=" if r.next_ref_id == r.ref_id else self._ref_name(r.next_ref_id))
            aux = r.aux_sam if r.aux_sam is not None else aux_bam_to_sam(r.aux_bam or b"")
            f = [r.qname, str(r.flag), rn, str(pos + 1), str(r.mapq), format_cigar(cigar), rnext, str(r.next_pos + 1),
                 str(r.tlen), r.seq if r.seq is not None else "*",
                 "".join(chr(q + 33) for q in r.qual) if r.qual is not None else "*"] + list(aux)
            self._f.write("\t".join(f) + "\n")

    def close(self):
        if self.mode == "wb":
            self._w.close()
        elif self._f is not sys.stdout:
            self._f.close()
