"""End to end on the host test double: FASTQ files -> arx_feeder -> the path -> arx_recbuf (AppendBam's record logic) -> arx_bam, the loop
of Arachne() (aligner.go:335-371, bamwriter.go:615-658) as arachne_amd/e2e.py drives it through the C ABI.  The BAM that comes out is
read back with an independent reader and every record is compared with what the rules of bamwriter.go:283-568 give for the placed
candidate of its read (flags, position, MAPQ, mate fields, template length, CIGAR op codes, strand of bases and qualities, tags) --
derived here in Python from the candidate records of a second run over the same reads.  GPU variant: tests marked gpu."""
import os
import struct
import subprocess
import tempfile
import zlib

import numpy as np
import pytest

from arachne_amd import api, e2e, synth

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim", "libarx_hostsim.so")
CODE = "=ACMGRSVTWYHKDBN"


def _read_bam(path):
    raw = open(path, "rb").read()
    data, o = b"", 0
    while o < len(raw):
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        data += zlib.decompress(raw[o + 18:o + bsize - 8], -15)
        o += bsize
    l_text = struct.unpack_from("<i", data, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", data, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", data, o)[0]; o += 4
        refs.append(data[o:o + l - 1].decode()); o += l + 4
    recs = []
    while o < len(data):
        bs = struct.unpack_from("<i", data, o)[0]; o += 4
        rid, pos, l_name, mapq, bn, n_cig, flag, l_seq, mrid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", data, o)
        q = o + 32
        name = data[q:q + l_name - 1].decode(); q += l_name
        cig = np.frombuffer(data, dtype="<u4", count=n_cig, offset=q).copy(); q += 4 * n_cig
        packed = data[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join(CODE[b >> 4] + CODE[b & 15] for b in packed)[:l_seq]
        qual = data[q:q + l_seq]; q += l_seq
        recs.append(dict(name=name, rid=rid, pos=pos, mapq=mapq, flag=flag, mrid=mrid, mpos=mpos, tlen=tlen, cigar=cig, seq=seq, qual=qual, aux=data[q:o + bs]))
        o += bs
    return refs, recs


def _expected(rs, cands, cand_off, alns, cigars, post, p, side, barcode, attach):
    """bamwriter.go:283-568 for the primary record of read 2p + side"""
    r = 2 * p + side
    act = lambda x: [i for i in range(cand_off[x], cand_off[x + 1]) if cands["active"][i]][0]
    c, m = cands[act(r)], cands[act(r ^ 1)]
    un = lambda a: a["pos"] == -1 or (not a["is_proper"] and a["score"] - 17 < 19)
    flag = 1 | (0x80 if side else 0x40)
    if c["is_proper"]: flag |= 2
    if un(m): flag |= 8
    elif m["reversed"]: flag |= 0x20
    if post["duplicate"][act(r)]: flag |= 0x400
    if un(c): flag |= 4
    if c["reversed"]: flag |= 0x10
    tl = 0
    if m["pos"] != -1 and c["rid"] == m["rid"] and (c["is_proper"] or m["score"] - 17 >= 19):
        tl = -int(c["aend"] - m["pos"]) if c["reversed"] else int(m["aend"] - c["pos"])
    L = int(rs.lens[r])
    s = rs.seqs[r][:L]
    seq = "".join("TGCAN"[x] for x in s[::-1]) if c["reversed"] else "".join("ACGTN"[x] for x in s)
    cg = np.zeros(0, np.uint32)
    if c["reg"] >= 0:
        a = alns[c["reg"]]
        w = cigars[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]
        cg = (w & ~np.uint32(15)) | np.array([0, 1, 2, 4, 5], np.uint32)[w & 15]
    rg = b"VX:i:1" if rs.valid[p] else b"VX:i:0"                    # RG is the last field of the R1 header, whatever it is (reader.go:144-153)
    aux = b"RGZ" + rg + b"\0" + b"ASi" + struct.pack("<i", int(c["score"])) + b"XMZ0\0" + b"AMZ" + (b"1" if c["active_molecule"] else b"0") + b"\0" + b"XTC\0"
    if attach and "-" in barcode:
        aux += b"BXZ" + barcode.encode() + b"\0" + b"VXC\x01"
    return dict(name="r%09d" % p, rid=-1 if un(c) else int(c["rid"]), pos=-1 if un(c) else int(c["pos"]), mapq=0 if un(c) else int(c["mapq"]), flag=flag,
                mrid=-1 if un(m) else int(m["rid"]), mpos=-1 if un(m) else int(m["pos"]), tlen=tl, cigar=cg, seq=seq, qual=bytes([40] * L), aux=aux)


def _e2e(lib_path, n_bc, ppb, workers, invalid_frac=0.25):
    g = synth.make_genome(15, [400000, 150000])
    rs = synth.make_reads(16, g, n_bc, ppb, invalid_frac=invalid_frac)
    rs.seqs[5] = np.random.default_rng(1).integers(0, 4, size=150)       # an unmappable read: the unmapped rules of AppendBam
    d = tempfile.mkdtemp(prefix="arx_e2e_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    api.index_build(fa, fa, lib_path=lib_path)
    po = rs.pair_offsets()
    cuts = [int(po[len(po) * k // workers]) for k in range(workers)] + [rs.n_pairs]     # whole barcode groups per file pair
    files = []
    for k in range(workers):
        f1, f2 = os.path.join(d, f"r1_{k}.fq"), os.path.join(d, f"r2_{k}.fq")
        synth.write_fastq_fast(rs, f1, f2, cuts[k], cuts[k + 1])
        files.append((f1, f2))
    ref = api.Reference(fa, lib_path=lib_path)
    try:
        st = e2e.run(ref, files, os.path.join(d, "out"), pairs_per_batch=max(50, rs.n_pairs // (3 * workers)), bam_threads=2, rec_threads=3, lib_path=lib_path)
        assert st["pairs"] == rs.n_pairs and st["records"] == 2 * rs.n_pairs and st["batches"] >= 2 * workers
        # the same reads as one batch: the candidate records every BAM record must follow from
        flags = [api.worth_running_rfa(rs.barcodes[b], int(po[b + 1] - po[b])) for b in range(len(po) - 1)]
        b = ref.batch(rs.seqs, rs.lens).run()
        out = b.fetch()
        c = b.rfa(po, flags)
        post = b.post()["post"]
        b.free()
        names = ref.contigs()[0]
        n_unmapped = 0
        for k in range(workers):
            refs, recs = _read_bam(os.path.join(d, f"out.{k}.bam"))
            assert refs == names and len(recs) == 2 * (cuts[k + 1] - cuts[k])
            for i, rec in enumerate(recs):
                p, side = cuts[k] + i // 2, i % 2
                bc = rs.barcodes[rs.barcode_id[p]]
                exp = _expected(rs, c["cands"], c["cand_off"], out["alns"], out["cigars"], post, p, side, bc, True)
                for key in exp:
                    same = np.array_equal(rec[key], exp[key]) if key == "cigar" else rec[key] == exp[key]
                    assert same, (k, i, key, rec[key], exp[key])
                n_unmapped += (rec["flag"] >> 2) & 1
        assert n_unmapped >= 1
        return st
    finally:
        ref.close()


def test_fastq_to_bam_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    _e2e(SIM, 6, 60, 2)


@pytest.mark.gpu
def test_fastq_to_bam_gpu(built):
    st = _e2e(api.LIB_PATH, 24, 500, 3)
    assert st["pairs_per_s"] > 0
