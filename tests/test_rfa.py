"""The Go half (candidate post-processing, RFA placement, MAPQ).

The reference holds no vectors for it and its Go code cannot run here (parity unpinned by the reference), so
this file pins the CPU restatement (oracle/arx_oracle_rfa.c) with hand-derived known answers from the closed-form constants
of aligner.go, and the device path (through the C ABI: host test double on CPU, libarachne_amd.so on the GPU) against the
restatement, bit for bit, on seeded workloads."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import parity
import rfadrv
import workloads
from arachne_amd import api, synth

SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libarx_hostsim.so")
L_PAC = 1_000_000


def _rows(cands):
    """cands: list per read of dicts(pos, rev, score, nm, cigar=[(op,len)...]) on one contig at offset 0 -> oracle input rows.
    A reverse-strand candidate with leftmost position P covers [P, P+150): aligner.go:1577-1582 with gobwa.go:351-363."""
    reg_off, regs, alns, cig = [0], [], [], []
    for read in cands:
        for c in read:
            if c["rev"]:
                cend = c["pos"] - 1          # pos = Alignment_end + 1
                re_ = 2 * L_PAC - 1 - cend
                rb = re_ - 150
            else:
                rb, re_ = c["pos"], c["pos"] + 150
            r = [0] * 20
            r[0], r[1], r[2], r[3], r[4], r[5], r[6] = rb, re_, 0, 150, 0, c["score"], c["score"]
            regs.append(r)
            cg = c.get("cigar", [(0, 150)])
            alns.append([0, 0, 0, int(c["rev"]), 0, 0, c.get("nm", 0), len(cg), len(cig), c["score"], 0, 0])
            cig += [(ln << 4) | op for op, ln in cg]
        reg_off.append(len(regs))
    return dict(reg_off=np.array(reg_off), regs=np.array(regs, dtype=np.int64).reshape(-1, 20), alns=np.array(alns, dtype=np.int64).reshape(-1, 12),
                cigars=np.array(cig + [0], dtype=np.uint32))


def _run(cands, do_rfa=True, penalty=-4):
    n = len(cands)
    out = rfadrv.oracle_rfa(_rows(cands), np.full(n, 150, dtype=np.int32), [0, n // 2], [do_rfa], L_PAC, [0], penalty=penalty)
    return out["cands"], out["cand_off"]


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    return built


@pytest.mark.parametrize("dist,expect", [(-36, 0), (-35, 1), (0, 1), (749, 1), (750, 0)])
def test_is_pair_window(dist, expect):
    """isPair (aligner.go:1032-1063): FR, reverse.pos - forward.pos in [-35, 750)."""
    c, _ = _run([[dict(pos=5000, rev=False, score=150)], [dict(pos=5000 + dist, rev=True, score=150)]], do_rfa=False)
    assert list(c[:, 13]) == [expect, expect]      # is_proper
    assert list(c[:, 12]) == [1, 1]                # both active


def test_same_strand_or_other_contig_is_never_a_pair():
    c, _ = _run([[dict(pos=5000, rev=False, score=150)], [dict(pos=5200, rev=False, score=150)]], do_rfa=False)
    assert list(c[:, 13]) == [0, 0]


def test_candidate_statistics_and_filter():
    """GetAlignments (aligner.go:1529-1627): mismatches = NM - indel bases, indels/soft clips counted per op,
    log_alignment_probability = -2 mm - 3 indel ops - 5 S ops - 0.5 S bases; candidates below best-17 leave the RFA lists."""
    r1 = [dict(pos=1000, rev=False, score=140, nm=5, cigar=[(3, 10), (0, 60), (1, 2), (0, 40), (2, 1), (0, 38)]),
          dict(pos=9000, rev=False, score=123, nm=0),       # 140 - 17: kept
          dict(pos=20000, rev=False, score=122, nm=0)]      # dropped
    c, off = _run([r1, [dict(pos=1200, rev=True, score=150)]], do_rfa=False)
    a = c[0]
    assert (a[7], a[8], a[9], a[10]) == (2, 2, 1, 10)        # mismatches, indel ops, S ops, S bases
    assert a[11] == -2 * (2 * 2 + 3 * 2 + 5 * 1) - 10       # lap2 = 2 * (-4 - 6 - 5 - 5) = -40
    assert list(c[:3, 17]) == [1, 1, 0]
    assert list(off) == [0, 3, 4]


def test_read_without_hits_gets_a_placeholder():
    c, off = _run([[], [dict(pos=1200, rev=True, score=150)]], do_rfa=False)
    assert list(off) == [0, 1, 2]
    assert (c[0, 0], c[0, 2], c[0, 5], c[0, 12]) == (-1, -1, -1, 1)   # reg, pos, contig "", still the read's active record


def _dense_molecule(n_pairs, start=100000, step=700):
    reads = []
    for i in range(n_pairs):
        p = start + i * step
        reads.append([dict(pos=p, rev=False, score=150)])
        reads.append([dict(pos=p + 200, rev=True, score=150)])
    return reads


def test_unique_pairs_form_one_active_molecule_with_mapq_60():
    c, _ = _run(_dense_molecule(6))
    assert (c[:, 15] == 0).all() and (c[:, 16] == 1).all()       # one molecule, active (12 alignments > 4, density 1)
    assert (c[:, 14] == 60).all() and (c[:, 12] == 1).all()


def test_active_molecule_needs_more_than_four_reads():
    c, _ = _run(_dense_molecule(2) + [[dict(pos=500000 + i, rev=bool(i & 1), score=150)] for i in range(6)][:0])
    assert (c[:, 16] == 0).all()                                  # 4 active alignments: not > 4 (aligner.go:1242)


def test_lone_placement_moves_into_the_dense_molecule_and_gets_mapq_30():
    """A pair with two equally good placements: the first-listed one far away (alone), the second inside a molecule of
    six other pairs.  tagBestAlignments keeps the first (exact tie -> first pair wins here); fastScore(lone -> dense) =
    +3 (the source molecule empties, aligner.go:1221-1226) > 0, so the sweep moves both reads.  Afterwards method 2 sees
    fastScore(dense -> lone) = -3 (sink empty): sum = 1 + 10^-3, MAPQ = int(-10 log10(1 - 1/1.001)) = 30."""
    reads = _dense_molecule(6)
    reads.append([dict(pos=800000, rev=False, score=150), dict(pos=102000, rev=False, score=150)])
    reads.append([dict(pos=800200, rev=True, score=150), dict(pos=102200, rev=True, score=150)])
    c, off = _run(reads)
    pair = c[off[12]:off[14]]
    assert list(pair[:, 12]) == [0, 1, 0, 1]                       # moved to the second-listed placement
    assert list(pair[pair[:, 12] == 1, 14]) == [30, 30]
    assert (c[:off[12], 14] == 60).all()
    # without RFA (barcode not worth it) the first-listed placement stays and both placements tie in method 1: -10 log10(1 - 1/(2 + pseudo)) ~ 3
    c2, off2 = _run(reads, do_rfa=False)
    pair2 = c2[off2[12]:off2[14]]
    assert list(pair2[:, 12]) == [1, 0, 1, 0]
    assert set(pair2[pair2[:, 12] == 1, 14]) == {3}


def test_worth_running_rfa_rule():
    assert rfadrv.worth_running_rfa("A01C02B03D04-1", 5)
    assert not rfadrv.worth_running_rfa("A01C02B03D04-1", 4)
    assert not rfadrv.worth_running_rfa("A01C02B03D04", 1000)
    assert not rfadrv.worth_running_rfa("A01C02B03D04-1", 1000, unique=False)


stats = {}


def _device_vs_oracle(lib_path, seed):
    import oradrv
    if seed % 2:
        g = synth.make_genome(31 + seed, [2000000, 600000])
        rs = synth.make_reads(32 + seed, g, 5, 300)
    else:
        g = workloads.nasty_genome(seed, contig_lens=(200000, 120000, 50000), alt_contigs=2)
        rs = workloads.nasty_reads(seed, g, n_barcodes=6, pairs_per_barcode=250)
    po = rs.pair_offsets()
    # PCR duplicates and unmappable pairs for markDuplicates: inside every barcode some pairs become copies of another pair,
    # some become random sequence (all their placeholders share one duplicate tuple, aligner.go:622-637)
    rng = np.random.default_rng(900 + seed)
    for bi in range(len(po) - 1):
        lo, hi = int(po[bi]), int(po[bi + 1])
        for _ in range(max(2, (hi - lo) // 20)):
            i, j = rng.integers(lo, hi, size=2)
            rs.seqs[2 * j:2 * j + 2] = rs.seqs[2 * i:2 * i + 2]
            rs.lens[2 * j:2 * j + 2] = rs.lens[2 * i:2 * i + 2]
        for j in rng.integers(lo, hi, size=3):
            rs.seqs[2 * j:2 * j + 2] = rng.integers(0, 4, size=rs.seqs[2 * j:2 * j + 2].shape)
    d = tempfile.mkdtemp(prefix="arx_rfa_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    g.write_alt(fa + ".alt")
    api.index_build(fa, fa, lib_path=lib_path)
    o = oradrv.Oracle(fa)
    ref = api.Reference(fa, lib_path=lib_path)
    flags = [rfadrv.worth_running_rfa(rs.barcodes[b], int(po[b + 1] - po[b])) for b in range(len(po) - 1)]
    flags[1] = False                                  # one barcode takes the non-RFA branch (aligner.go:469-477)
    names, offs, clens, alt, l_pac = ref.contigs()
    cen = (np.array(offs) * 0 + 1000, np.array(offs) * 0 + 30000)   # a "centromere" on every contig: MAPQ forced to 0 inside
    ob = o.batch(rs.seqs, rs.lens, n_threads=4)
    ora = rfadrv.oracle_rfa(ob, rs.lens, po, flags, l_pac, offs, centromeres=cen)
    b = ref.batch(rs.seqs, rs.lens).run()
    dev = b.rfa(po, flags, centromeres=cen)
    parity.check_rfa(dev, ora)
    # the passes between placement and the BAM records: CIGAR walk, markDuplicates, CheckSplitReads
    opost = rfadrv.oracle_post(o.h, ob, rs.seqs, rs.lens, po, offs, ora, centromeres=cen)
    dpost = b.post()
    parity.check_post(dpost, opost)
    stats["n_mm"] = stats.get("n_mm", 0) + len(dpost["mm_ref"])
    stats["dups"] = stats.get("dups", 0) + int(dpost["post"]["duplicate"].sum())
    stats["splits"] = stats.get("splits", 0) + int((dpost["split"]["split"] >= 0).sum())
    stats["unpinned"] = stats.get("unpinned", 0) + int((dpost["split"]["order_pinned"] == 0).sum())
    act = dev["cands"][dev["cands"]["active"] == 1]
    assert len(act) == len(rs.lens)                   # exactly one active candidate per read
    b.free()
    ref.close()


@pytest.mark.parametrize("seed,lane_order", [(1, 0), (2, 0), (2, 1), (1, 2), (2, 2)])
def test_device_logic_matches_restatement_hostsim(seed, lane_order, monkeypatch):
    """lane_order != 0 runs every workgroup-parallel phase of the RFA code in another lane order: the result may not depend on it."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    monkeypatch.setenv("ARX_SIM_PFOR", str(lane_order))
    _device_vs_oracle(SIM, seed)


def test_host_mapq_path_hostsim(monkeypatch):
    """ARX_MAPQ_GUARD=0.6 sends every read through the host re-evaluation + patch path (normally a handful per batch)."""
    monkeypatch.setenv("ARX_MAPQ_GUARD", "0.6")
    _device_vs_oracle(SIM, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,guard", [(3, None), (4, None), (4, "0.6")])
def test_gpu_matches_restatement(seed, guard, monkeypatch):
    """MAPQ comes from the device's pow/log10 except within the guard band of an integer, where the host's libm decides;
    guard=0.6 forces the host path for every read."""
    if guard:
        monkeypatch.setenv("ARX_MAPQ_GUARD", guard)
    _device_vs_oracle(api.LIB_PATH, seed)


def test_non_integer_penalty_is_rejected_with_a_message_hostsim():
    """The reference's -i flag is a float64 (main.go:28); the path keeps scores in exact half-units, so arx_batch_rfa takes the value as
    a double and refuses a non-integer one (ARX_E_ARG + message) instead of rounding it silently; integers given as floats pass."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    g = synth.make_genome(5, [200000])
    rs = synth.make_reads(6, g, 1, 12)
    d = tempfile.mkdtemp(prefix="arx_pen_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    api.index_build(fa, fa, lib_path=SIM)
    ref = api.Reference(fa, lib_path=SIM)
    b = ref.batch(rs.seqs, rs.lens).run()
    with pytest.raises(api.ArachneError, match="must be an integer"):
        b.rfa([0, 12], [True], penalty=-4.5)
    with pytest.raises(api.ArachneError, match="must be an integer"):
        b.rfa([0, 12], [True], penalty=float("nan"))
    a = b.rfa([0, 12], [True], penalty=-4.0)
    c = b.rfa([0, 12], [True], penalty=-4)
    assert a["cands"].tobytes() == c["cands"].tobytes()
    b.free()
    ref.close()
