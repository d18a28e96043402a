"""The BAM sink (arx_bam_*): what reaches the file is, after BGZF decompression, exactly the SAM/BAM-specification
encoding of the records handed in -- checked by an independent reader written here (Python's zlib per BGZF block + struct): header,
references, every fixed field, bin (reg2bin), CIGAR, 4-bit bases, qualities, aux bytes; BGZF framing (BC extra field, BSIZE, CRC32, ISIZE,
blocks of at most 64 KiB, the 28-byte EOF marker); the same bytes whatever the number of threads.  The reference's writer (biogo/hts
v1.4.5, go.mod:5) is not in /root/reference: parity with ITS bytes is unpinned; the specification is the pin."""
import os
import struct
import zlib

import numpy as np

from arachne_amd import api


def _bgzf_blocks(raw):
    out, o = [], 0
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04", o
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        assert xlen == 6 and raw[o + 12:o + 14] == b"BC" and struct.unpack_from("<H", raw, o + 14)[0] == 2
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        assert bsize <= 65536
        body = raw[o + 18:o + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, o + bsize - 8)
        data = zlib.decompress(body, -15)
        assert len(data) == isize and zlib.crc32(data) == crc and isize <= 65280
        out.append(data)
        o += bsize
    return out


def _reg2bin(beg, end):
    end -= 1
    for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return base + (beg >> sh)
    return 0


def _records(n, rng, names_c):
    recs = []
    for i in range(n):
        l_seq = int(rng.integers(0, 200)) if i % 17 else 0
        unm = i % 11 == 0
        ops = [(int(rng.integers(0, 9)), int(rng.integers(1, 90))) for _ in range(int(rng.integers(0, 6)))] if not unm else []
        cig = np.array([l << 4 | op for op, l in ops], dtype=np.uint32)
        name = ("r%d:%s" % (i, "x" * int(rng.integers(0, 40)))).encode()
        seq = bytes(rng.choice(list(b"ACGTNacgtn=MRSVWYHKDB"), size=l_seq).astype(np.uint8))
        qual = bytes((rng.integers(0, 42, size=l_seq) + 33).astype(np.uint8))
        aux = b"ASC" + bytes([int(rng.integers(0, 150))]) + b"BXZ" + ("A%02dC%02d-1" % (i % 96, i % 7)).encode() + b"\0" if i % 3 else b""
        recs.append(dict(name=name, flag=int(rng.integers(0, 4096)), rid=-1 if unm else int(rng.integers(0, names_c)), pos=-1 if unm else int(rng.integers(0, 2 ** 29 - 5000)),
                         mapq=int(rng.integers(0, 61)), mate_rid=int(rng.integers(-1, names_c)), mate_pos=int(rng.integers(-1, 1000000)), tlen=int(rng.integers(-900, 900)),
                         cigar=cig, seq=seq, qual=qual, aux=aux))
    return recs


def _write(path, recs, threads, batch):
    w = api.BamWriter(path, ["chrA", "chrB_random", "c3"], [600000000, 1234567, 88], extra_header="@RG\tID:lib1\tSM:s\n@PG\tID:arachne_amd\n", threads=threads)
    for o in range(0, len(recs), batch):
        part = recs[o:o + batch]
        w.write([r["name"] for r in part], [r["flag"] for r in part], [r["rid"] for r in part], [r["pos"] for r in part], [r["mapq"] for r in part],
                [r["mate_rid"] for r in part], [r["mate_pos"] for r in part], [r["tlen"] for r in part], [r["cigar"] for r in part],
                [r["seq"] for r in part], [r["qual"] for r in part], [r["aux"] for r in part])
    return w.close()


def test_bam_stream_is_the_specified_encoding(tmp_path):
    rng = np.random.default_rng(3)
    recs = _records(5000, rng, 3)
    p1, p8 = str(tmp_path / "a1.bam"), str(tmp_path / "a8.bam")
    st = _write(p1, recs, 1, 700)
    _write(p8, recs, 8, 1999)
    raw = open(p1, "rb").read()
    assert raw == open(p8, "rb").read()                       # threads and batch size do not change a byte
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    blocks = _bgzf_blocks(raw)
    assert blocks[-1] == b"" and st["records"] == 5000 and st["blocks"] == len(blocks) - 1 and st["bytes_out"] == len(raw)
    data = b"".join(blocks)
    assert st["bytes_in"] == len(data)
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    assert text.startswith("@HD\tVN:1.6") and "@SQ\tSN:chrB_random\tLN:1234567\n" in text and text.endswith("@PG\tID:arachne_amd\n")
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", data, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", data, o)[0]; o += 4
        refs.append((data[o:o + l - 1].decode(), struct.unpack_from("<i", data, o + l)[0])); assert data[o + l - 1] == 0
        o += l + 4
    assert refs == [("chrA", 600000000), ("chrB_random", 1234567), ("c3", 88)]
    code = "=ACMGRSVTWYHKDBN"
    for r in recs:
        bs = struct.unpack_from("<i", data, o)[0]; o += 4
        rid, pos, l_name, mapq, bn, n_cig, flag, l_seq, mrid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", data, o)
        q = o + 32
        assert (rid, pos, mapq, flag, mrid, mpos, tlen) == (r["rid"], r["pos"], r["mapq"], r["flag"], r["mate_rid"], r["mate_pos"], r["tlen"])
        assert data[q:q + l_name] == r["name"] + b"\0"; q += l_name
        cg = np.frombuffer(data, dtype="<u4", count=n_cig, offset=q); q += 4 * n_cig
        assert np.array_equal(cg, r["cigar"])
        ref_len = sum(int(w) >> 4 for w in r["cigar"] if (int(w) & 15) in (0, 2, 3, 7, 8))
        assert bn == (4680 if r["pos"] < 0 else _reg2bin(r["pos"], r["pos"] + max(ref_len, 1)))
        assert l_seq == len(r["seq"])
        packed = data[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        dec = "".join(code[b >> 4] + code[b & 15] for b in packed)[:l_seq]
        exp = "".join(c.upper() if c.upper() in code else "N" for c in r["seq"].decode())
        assert dec == exp
        assert data[q:q + l_seq] == bytes(b - 33 for b in r["qual"]); q += l_seq
        assert data[q:o + bs] == r["aux"]
        o += bs
    assert o == len(data)


def test_bam_open_reports_errors(tmp_path):
    import pytest
    with pytest.raises(api.ArachneError):
        api.BamWriter(str(tmp_path / "no_such_dir" / "x.bam"), ["c"], [10])
    w = api.BamWriter(str(tmp_path / "e.bam"), ["c"], [10])
    with pytest.raises(api.ArachneError):
        w.write([b""], [0], [0], [0], [0], [-1], [-1], [0], [np.zeros(0, np.uint32)], [b"A"], [b"I"], [b""])     # empty read name
    w.close()
