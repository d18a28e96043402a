"""The FASTQ feeder in front of the path: arx_feeder_* of the product library against the plain-Python
restatement of the reference's reader (oracle/fastq_reader.py: ParseHeader, ReadOneLine with the intended 4-line semantics,
ReadBarcodeSet, worthRunningRFA), and both against set sizes and flags derived by hand from reader.go:209-300.
The reference holds no vectors for its reader: parity unpinned by the reference, pinned by the hand-derived cases here."""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import fastq_reader  # noqa: E402

from arachne_amd import api, synth  # noqa: E402

SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libarx_hostsim.so")
ACGT = "ACGTN"


def _fastq(groups, seed=0, read_len=40):
    """groups: list of (barcode or None, n_records) -> (r1_text, r2_text)"""
    rng = np.random.default_rng(seed)
    r1, r2 = [], []
    k = 0
    for bc, n in groups:
        for _ in range(n):
            s1 = "".join(ACGT[x] for x in rng.integers(0, 5, size=read_len))
            s2 = "".join(ACGT[x] for x in rng.integers(0, 4, size=read_len - 3))
            q1 = "".join(chr(33 + x) for x in rng.integers(0, 40, size=read_len))
            q2 = "".join(chr(33 + x) for x in rng.integers(0, 40, size=read_len - 3))
            tags = "" if bc is None else f"\tBX:Z:{bc}\tVX:i:{k % 2}"
            r1.append(f"@read{k}/1{tags}\n{s1}\n+\n{q1}\n")
            r2.append(f"@read{k}/2{tags}\n{s2}\n+\n{q2}\n")
            k += 1
    return "".join(r1), "".join(r2)


def _write(d, name, text, gz=False):
    p = os.path.join(d, name)
    if gz:
        with gzip.open(p, "wt", newline="\n") as f:
            f.write(text)
    else:
        with open(p, "w", newline="\n") as f:
            f.write(text)
    return p


def _feed_all(r1, r2, target, lib_path):
    fd = api.Feeder(r1, r2, lib_path=lib_path)
    out = []
    while True:
        sb = fd.next(target)
        if sb is None:
            break
        out.append(sb)
    assert fd.next(target) is None            # stays at the end
    fd.close()
    return out


def _check_against_restatement(batches, r1_text, r2_text):
    sets, bad = fastq_reader.all_sets(r1_text, r2_text)
    k = 0
    for sb in batches:
        boff = np.concatenate([[0], np.cumsum(sb["lens"])])
        for s in range(sb["n_sets"]):
            recs, unique, rfa = sets[k]
            k += 1
            p0, p1 = int(sb["set_pair_off"][s]), int(sb["set_pair_off"][s + 1])
            assert p1 - p0 == len(recs)
            assert (bool(sb["unique"][s]), bool(sb["do_rfa"][s])) == (unique, rfa)
            assert sb["barcodes"][s] == recs[0]["barcode"]
            for j, rec in enumerate(recs):
                p = p0 + j
                assert sb["names"][p] == rec["info"] and sb["rgs"][p] == rec["rg"] and bool(sb["valid"][p]) == rec["valid"]
                for side, (sq, ql) in enumerate(((rec["s1"], rec["q1"]), (rec["s2"], rec["q2"]))):
                    a, b = boff[2 * p + side], boff[2 * p + side + 1]
                    assert b - a == len(sq)
                    assert "".join(ACGT[x] for x in sb["bases"][a:b]) == sq.upper().translate(str.maketrans("RYKMSWBDHVN", "N" * 11))
                    assert sb["quals"][a:b].decode() == ql
    assert k == len(sets)
    assert batches[-1]["bad_lines"] == bad
    return sets


@pytest.fixture(scope="module")
def lib(built):
    return api.LIB_PATH


def test_barcode_set_rules(lib):
    """Hand-derived from reader.go:209-300.  A-1 x3: unique, too small for RFA (< 5).  B-1 x5: RFA.  C x7: no '-': no RFA.
    D-1 x30450: 30000 (cap: not unique), then 201 + 201 ("abnormal break" at index 200 keeps the 201st record: not unique), then the
    48 left end at the barcode change: unique again, RFA.  E-1 x1.  Two records without BX: barcode "", empty ReadInfo.
    F-1 x6 with the last line of the file unterminated: ReadString reports EOF inside that record and it is dropped -> 5, RFA."""
    groups = [("A-1", 3), ("B-1", 5), ("C", 7), ("D-1", 30450), ("E-1", 1), (None, 2), ("F-1", 6)]
    t1, t2 = _fastq(groups)
    t1, t2 = t1[:-1], t2[:-1]                   # no newline at the end of the files
    d = tempfile.mkdtemp(prefix="arx_feed_")
    batches = _feed_all(_write(d, "r1.fq", t1), _write(d, "r2.fq", t2), 1000, lib)
    sizes = [int(x) for sb in batches for x in np.diff(sb["set_pair_off"])]
    unique = [int(x) for sb in batches for x in sb["unique"]]
    rfa = [int(x) for sb in batches for x in sb["do_rfa"]]
    assert sizes == [3, 5, 7, 30000, 201, 201, 48, 1, 2, 5]
    assert unique == [1, 1, 1, 0, 0, 0, 1, 1, 1, 1]
    assert rfa == [0, 1, 0, 0, 0, 0, 1, 0, 0, 1]
    assert [sb["n_sets"] for sb in batches] == [4, 6]      # whole sets until >= 1000 pairs
    assert batches[1]["barcodes"][4] == "" and batches[1]["names"][201 + 201 + 48 + 1] == ""
    assert batches[0]["names"][0] == "read0" and batches[0]["rgs"][0] == "VX:i:0"      # ReadInfo minus "/1"; last header field
    assert list(batches[0]["valid"][:4]) == [0, 1, 0, 1]
    _check_against_restatement(batches, t1, t2)


def test_gzip_bad_lines_and_header_forms(lib):
    """gzip input; a stray line between records is skipped in both files (reader.go:156-159); BX at the end of the header line, BX
    followed by nothing, several BX tags (leftmost wins), VX:i:2 (no match: not valid), a header of one field (empty ReadGroupId)."""
    r1 = ("@a/1 BX:Z:X-1\nACGT\n+\nIIII\n" "stray\n" "@b/1\tVX:i:1\tBX:Z:X-1\tBX:Z:Y-1\nacgn\n+\nIIII\n" "@c/1 BX:Z: VX:i:1\nAC\n+\nII\n"
          "@d/1\nA\n+\nI\n" "@e/1 BX:Z:Z-1 VX:i:2\nA\n+\nI\n")
    r2 = ("@a/2 BX:Z:X-1\nTTTT\n+\nJJJJ\n" "stray\n" "@b/2\nGGGG\n+\nJJJJ\n" "@c/2\nGG\n+\nJJ\n" "@d/2\nG\n+\nJ\n" "@e/2\nG\n+\nJ\n")
    d = tempfile.mkdtemp(prefix="arx_feed_")
    batches = _feed_all(_write(d, "r1.fq.gz", r1, gz=True), _write(d, "r2.fq.gz", r2, gz=True), 10, lib)
    sets = _check_against_restatement(batches, r1, r2)
    sb = batches[0]
    assert [len(s[0]) for s in sets] == [2, 2, 1]
    assert sb["barcodes"] == ["X-1", "", "Z-1"] and sb["bad_lines"] == 1
    assert sb["names"] == ["a", "b", "", "", "e"] and sb["rgs"] == ["BX:Z:X-1", "BX:Z:Y-1", "VX:i:1", "", "VX:i:2"]
    assert list(sb["valid"]) == [0, 1, 0, 0, 0]
    assert list(sb["bases"][4:8]) == [3, 3, 3, 3] and list(sb["bases"][8:12]) == [0, 1, 2, 4]      # nst_nt4_table: lower case maps too


def test_missing_file_is_an_error(lib):
    with pytest.raises(api.ArachneError):
        api.Feeder("/nonexistent/r1.fq", "/nonexistent/r2.fq", lib_path=lib)


def test_fastq_to_placements_hostsim(built):
    """End to end on the host test double: FASTQ -> feeder -> batch -> RFA gives what the arrays it was written from give."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    _fastq_to_placements(SIM, 4, 60)


@pytest.mark.gpu
def test_fastq_to_placements_gpu(built):
    """The same through libarachne_amd.so on the GPU, gzip input, two super-batches."""
    _fastq_to_placements(api.LIB_PATH, 12, 400, gz=True, target=3000)


def _fastq_to_placements(SIM, n_bc, ppb, gz=False, target=10**6):
    g = synth.make_genome(5, [300000, 100000])
    rs = synth.make_reads(6, g, n_bc, ppb)
    d = tempfile.mkdtemp(prefix="arx_feed_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    g.write_alt(fa + ".alt")
    api.index_build(fa, fa, lib_path=SIM)
    po = rs.pair_offsets()
    r1, r2 = [], []
    for b in range(len(po) - 1):
        for p in range(int(po[b]), int(po[b + 1])):
            for side, out in ((0, r1), (1, r2)):
                s = "".join(ACGT[x] for x in rs.seqs[2 * p + side][:rs.lens[2 * p + side]])
                out.append(f"@p{p}/{side + 1} BX:Z:{rs.barcodes[b]} VX:i:1\n{s}\n+\n{'I' * len(s)}\n")
    ext = ".fq.gz" if gz else ".fq"
    fd = api.Feeder(_write(d, "r1" + ext, "".join(r1), gz=gz), _write(d, "r2" + ext, "".join(r2), gz=gz), lib_path=SIM)
    ref = api.Reference(fa, lib_path=SIM)
    flags = [api.worth_running_rfa(rs.barcodes[b], int(po[b + 1] - po[b])) for b in range(len(po) - 1)]
    s0 = 0                                         # barcode sets consumed so far
    n_batches = 0
    while True:
        sb = fd.next(target)
        if sb is None:
            break
        n_batches += 1
        s1 = s0 + sb["n_sets"]
        assert (sb["set_pair_off"] == po[s0:s1 + 1] - po[s0]).all() and sb["barcodes"] == list(rs.barcodes[s0:s1])
        assert list(sb["do_rfa"]) == [int(x) for x in flags[s0:s1]]
        p0, p1 = int(po[s0]), int(po[s1])
        a = ref.batch(sb["bases"], sb["lens"]).run()
        b = ref.batch(rs.seqs[2 * p0:2 * p1], rs.lens[2 * p0:2 * p1]).run()
        ca, cb = a.rfa(sb["set_pair_off"], sb["do_rfa"]), b.rfa(po[s0:s1 + 1] - po[s0], flags[s0:s1])
        assert (ca["cand_off"] == cb["cand_off"]).all() and ca["cands"].tobytes() == cb["cands"].tobytes()
        pa, pb = a.post(), b.post()
        assert pa["post"].tobytes() == pb["post"].tobytes() and pa["split"].tobytes() == pb["split"].tobytes()
        a.free()
        b.free()
        s0 = s1
    assert s0 == len(po) - 1 and n_batches >= (2 if target < rs.n_pairs else 1)
    ref.close()
