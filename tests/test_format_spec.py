"""The file-format surface against the published specifications, without htslib:
known-answer vectors from the SAM/BAM specification (SAMv1 section 4.1 BGZF, 4.2 BAM records, 5.3 reg2bin) and a
parser written here from the specification text alone -- it shares no code with amplipy_amd/bamio.py or libampbam --
cross-reading files produced by BOTH writers of the package; the VCF header field by field against the keys the
reference adds (AmpliPy.py:271-281).  No GPU needed."""
import gzip
import io
import os
import re
import struct
import zlib

import pytest

from amplipy_amd import bam_native, bamio

# SAMv1 section 4.1.2: "an end-of-file (EOF) marker ... a valid BGZF block containing no data", these 28 bytes
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def spec_reg2bin(beg, end):
    """SAMv1 section 5.3, C code of the specification transcribed: bin of the 0-based half-open interval [beg, end)."""
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def spec_parse_bam(path):
    """(header text, [(name, length)], [record dict]) of a BAM file, from SAMv1 sections 4.1 and 4.2 only."""
    raw = open(path, "rb").read()
    assert raw.endswith(BGZF_EOF), "no BGZF end-of-file marker"
    data, off = bytearray(), 0
    while off < len(raw):                                              # 4.1: a series of gzip members with a BC extra field
        id1, id2, cm, flg, _mt, _xfl, _os, xlen = struct.unpack_from("<BBBBIBBH", raw, off)
        assert (id1, id2, cm, flg) == (31, 139, 8, 4)
        extra = raw[off + 12:off + 12 + xlen]
        assert extra[:4] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", extra, 4)[0] + 1
        cdata = raw[off + 12 + xlen:off + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, off + bsize - 8)
        block = zlib.decompress(cdata, -15)
        assert len(block) == isize and zlib.crc32(block) == crc
        data += block
        off += bsize
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].rstrip(b"\x00").decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]; p += 4
        name = data[p:p + l_name - 1].decode(); p += l_name
        refs.append((name, struct.unpack_from("<i", data, p)[0])); p += 4
    recs = []
    while p < len(data):
        block_size = struct.unpack_from("<i", data, p)[0]; p += 4
        ref_id, pos, l_read_name, mapq, bin_, n_cigar_op, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHiiii", data, p)
        q = p + 32
        name = data[q:q + l_read_name - 1].decode(); q += l_read_name
        cigar = [(w & 15, w >> 4) for w in struct.unpack_from("<%dI" % n_cigar_op, data, q)]; q += 4 * n_cigar_op
        seq = "".join("=ACMGRSVTWYHKDBN"[(data[q + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq)); q += (l_seq + 1) // 2
        qual = bytes(data[q:q + l_seq]); q += l_seq
        recs.append(dict(name=name, flag=flag, ref_id=ref_id, pos=pos, mapq=mapq, bin=bin_, cigar=cigar, next_ref=next_ref,
                         next_pos=next_pos, tlen=tlen, seq=seq, qual=qual, aux=bytes(data[q:p + block_size])))
        p += block_size
    return text, refs, recs


def test_reg2bin_known_answers():
    # first bins of the six levels (SAMv1 section 5.3: 0, 1-8, 9-72, 73-584, 585-4680, 4681-37448) and boundary cases
    for beg, end, want in [(0, 1, 4681), (0, 1 << 14, 4681), (0, (1 << 14) + 1, 585), (1 << 14, (1 << 14) + 1, 4682),
                           (0, 1 << 17, 585), (0, (1 << 17) + 1, 73), (0, (1 << 20) + 1, 9), (0, (1 << 23) + 1, 1),
                           (0, (1 << 26) + 1, 0), ((1 << 29) - 1, 1 << 29, 37448), (29902, 29903, 4682), (16383, 16385, 585)]:
        assert spec_reg2bin(beg, end) == want
        assert bamio.reg2bin(beg, end) == want, (beg, end)


def _records():
    return [
        bamio.Rec("r1", 99, 0, 25, 60, [(0, 10)], 0, 300, 285, "ACGTACGTAC", bytes(range(30, 40)), aux_sam=["NM:i:1"]),
        bamio.Rec("r2", 147, 0, 16380, 37, [(4, 2), (0, 6), (1, 1), (0, 3), (2, 2), (0, 4)], 0, 25, -285, "NNACGTACGTACGTAC", bytes([20] * 16)),
        bamio.Rec("r3", 0, 0, 29890, 0, [(0, 13)], -1, -1, 0, "GATTACAGATTAC", None),                    # QUAL '*'
        bamio.Rec("r4", 4, -1, -1, 0, None, -1, -1, 0, "ACG", bytes([2, 3, 4])),                          # unmapped, no CIGAR
    ]


@pytest.mark.parametrize("writer", ["python", "native"])
def test_written_bam_reads_back_with_a_spec_level_parser(tmp_path, writer):
    hdr = bamio.Header("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:REF\tLN:29903\n", [("REF", 29903)])
    recs = _records()
    path = str(tmp_path / "t.bam")
    if writer == "python":
        w = bamio.AlignmentWriter(path, "wb", hdr)
        for r in recs:
            w.write(r)
        w.close()
    else:
        # libampbam re-encodes the rows of an input BAM: feed it the Python writer's file, rows unchanged
        src = str(tmp_path / "src.bam")
        w = bamio.AlignmentWriter(src, "wb", hdr)
        for r in recs:
            w.write(r)
        w.close()
        f = bam_native.BamFile(src, threads=2)
        batch, _ = f.decode(0, f.n_records, copy=True)
        import numpy as np
        out = bam_native.BamWriter(path, hdr.text, f, level=6, threads=2)
        out.write_rows(None, batch.src_index, np.ones(batch.n, np.uint8), batch.pos, np.diff(batch.cig_off), batch.cig_off[:-1], batch.cig)
        out.close()
        recs = [r for r in recs if not (r.flag & 4) and r.cigar is not None]     # the rows the reference's loop keeps (A:902)
    text, refs, got = spec_parse_bam(path)
    assert text == hdr.text and refs == [("REF", 29903)]
    assert len(got) == len(recs)
    for g, r in zip(got, recs):
        assert (g["name"], g["flag"], g["ref_id"], g["pos"], g["mapq"]) == (r.qname, r.flag, r.ref_id, r.pos, r.mapq)
        assert g["cigar"] == [tuple(c) for c in (r.cigar or [])]
        assert g["seq"] == r.seq and (g["qual"] == (r.qual if r.qual is not None else b"\xff" * len(r.seq)))
        assert (g["next_ref"], g["next_pos"], g["tlen"]) == (r.next_ref_id, r.next_pos, r.tlen)
        if r.cigar is not None and not (r.flag & 4):
            ref_len = sum(l for op, l in r.cigar if op in (0, 2, 3, 7, 8)) or 1
            assert g["bin"] == spec_reg2bin(r.pos, r.pos + ref_len)
    # every member also opens with Python's gzip module (BGZF is a gzip file)
    assert gzip.open(path, "rb").read()[:4] == b"BAM\x01"


def test_vcf_header_lines_field_by_field(tmp_path, monkeypatch):
    """AmpliPy.py:271-281: sample 'sample'; meta AmpliPyVersion, source, contig; FORMAT GT; INFO DP, REF_DP, ALT_DP, REF_FREQ,
    ALT_FREQ with these Number / Type / Description values."""
    from amplipy_amd import amplipy
    monkeypatch.setattr("sys.argv", ["AmpliPy.py", "variants", "-i", "x.bam"])
    fn = str(tmp_path / "v.vcf")
    w = amplipy.VcfWriter(fn, "REF_ID")
    w.close()
    lines = open(fn).read().splitlines()
    assert lines[0] == "##fileformat=VCFv4.2"
    meta = {}
    for ln in lines:
        m = re.match(r"##(\w+)=<(.*)>$", ln)
        if m:
            fields = dict(kv.split("=", 1) for kv in re.findall(r'(\w+=(?:"[^"]*"|[^,]*))', m.group(2)))
            meta.setdefault(m.group(1), {})[fields["ID"]] = fields
    assert "##AmpliPyVersion=%s" % amplipy.VERSION in lines
    assert "##source=AmpliPy.py variants -i x.bam" in lines
    assert meta["contig"] == {"REF_ID": {"ID": "REF_ID"}}
    assert meta["FORMAT"]["GT"] == {"ID": "GT", "Number": "1", "Type": "String", "Description": '"Genotype"'}
    want = {"DP": ("1", "Integer", "Total Depth"), "REF_DP": ("1", "Integer", "Depth of reference base"),
            "ALT_DP": ("1", "String", "Depth of alternate base"), "REF_FREQ": ("1", "Float", "Frequency of reference base"),
            "ALT_FREQ": ("1", "String", "Frequency of alternate base")}
    assert set(meta["INFO"]) == set(want)
    for k, (num, typ, desc) in want.items():
        assert meta["INFO"][k] == {"ID": k, "Number": num, "Type": typ, "Description": '"%s"' % desc}
    assert lines[-1].split("\t") == ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT", "sample"]
