"""The passes between placement and the BAM records: GetAlignments' CIGAR walk with mismatch locations
(aligner.go:1505-1570), markDuplicates (aligner.go:611), CheckSplitReads (split.go:144).

The reference holds no vectors for its Go half and the Go tree cannot run here (parity unpinned by the reference), so the CPU
restatement (oracle/arx_oracle_rfa.c: ora_post) is pinned by known answers derived by hand from the reference's code on reads
with planted differences, and the device path (C ABI; host test double here, libarachne_amd.so under -m gpu) is compared with the
restatement record by record here and on the seeded workloads of test_rfa.py."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oradrv
import parity
import rfadrv
from arachne_amd import api, synth

SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libarx_hostsim.so")
RL = 150


def _rc(a):
    return (3 - a[::-1]).astype(np.uint8)


class Case:
    """A unique random genome (no repeats) and hand-made pairs on contig 1 (offset 400000 in the concatenation)."""

    def __init__(self, lib_path):
        self.g = synth.make_genome(77, [400000, 300000], repeat_families=[])
        for s in self.g.seqs:
            s[s > 3] = 0                              # no N runs: every reference base is what the test planted against
        d = tempfile.mkdtemp(prefix="arx_post_")
        self.fa = os.path.join(d, "g.fa")
        self.g.write_fasta(self.fa)
        self.g.write_alt(self.fa + ".alt")
        api.index_build(self.fa, self.fa, lib_path=lib_path)
        self.lib_path = lib_path
        self.c1 = self.g.seqs[1]

    def fwd(self, pos, subs=()):
        r = self.c1[pos:pos + RL].copy()
        for x in subs:
            r[x] = (r[x] + 1) & 3
        return r

    def rev(self, pos, subs=()):
        """the read is the reverse complement of [pos, pos+150); subs are offsets in the read as sequenced"""
        r = _rc(self.c1[pos:pos + RL])
        for x in subs:
            r[x] = (r[x] + 1) & 3
        return r

    def run(self, pairs, centromeres=None):
        seqs = np.concatenate([np.concatenate(p) for p in pairs])
        lens = np.full(2 * len(pairs), RL, dtype=np.int32)
        po = [0, len(pairs)]
        o = oradrv.Oracle(self.fa)
        ref = api.Reference(self.fa, lib_path=self.lib_path)
        names, offs, clens, alt, l_pac = ref.contigs()
        ob = o.batch(seqs, lens)
        ora = rfadrv.oracle_rfa(ob, lens, po, [False], l_pac, offs, centromeres=centromeres)
        opost = rfadrv.oracle_post(o.h, ob, seqs, lens, po, offs, ora, centromeres=centromeres)
        b = ref.batch(seqs, lens).run()
        dev = b.rfa(po, [False], centromeres=centromeres)
        parity.check_rfa(dev, ora)
        parity.check_post(b.post(), opost)            # the device path says the same, record by record
        b.free()
        ref.close()
        return ora, opost


@pytest.fixture(scope="module")
def case(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    return Case(SIM)


def _mm(post, i):
    p = post["post"][i]
    return list(post["mm_ref"][p[4]:p[4] + p[3]]), list(post["mm_read"][p[4]:p[4] + p[3]])


def test_walk_forward_and_reverse(case):
    """Forward read at P with substitutions at read offsets 10 and 77: mismatchLocs = P+10, P+77 (aligner.go:1546),
    mismatchReadLocs = 10, 77.  Reverse read over [Q, Q+150) with a substitution at read offset 20: the walk runs over the
    reverse-complemented reference string, mismatchLocs = refEnd - 20 = Q+150-20 (aligner.go:1544) -- one past the base itself,
    as the reference computes it -- and mismatchReadLocs = 20 in the read as sequenced."""
    P, Q = 50000, 50250
    ora, post = case.run([(case.fwd(P, [10, 77]), case.rev(Q, [20]))])
    c = ora["cands"]
    assert len(c) == 2 and list(c[:, 2]) == [P, Q]                   # Alignment.pos of both
    assert list(post["post"][0][:4]) == [0, RL, RL - 2, 2]          # qb, qe, matches = 150 - NM, two locations
    assert _mm(post, 0) == ([P + 10, P + 77], [10, 77])
    assert list(post["post"][1][:4]) == [0, RL, RL - 1, 1]
    assert _mm(post, 1) == ([Q + RL - 20], [20])
    assert list(post["post"][:, 5]) == [0, 0]
    assert list(post["split"][:, 0]) == [-1, -1]                     # fully aligned reads are never split (split.go:48-50)


def test_duplicates_first_pair_wins(case):
    """markDuplicates (aligner.go:611-641): pairs 0, 2, 3 are the same fragment (pair 3 with other sequencing errors, same
    positions), pair 1 another one, pairs 4 and 5 random sequence (all four placeholders share (contig "", pos -1) per read1 flag)."""
    rng = np.random.default_rng(5)
    P, Q = 120000, 120300
    a = (case.fwd(P), case.rev(Q))
    junk = lambda: rng.integers(0, 4, size=RL).astype(np.uint8)  # noqa: E731
    pairs = [a, (case.fwd(P + 1000), case.rev(Q + 1000)), a, (case.fwd(P, [5]), case.rev(Q, [9])), (junk(), junk()), (junk(), junk())]
    ora, post = case.run(pairs)
    act = [int(np.flatnonzero(ora["cands"][ora["cand_off"][r]:ora["cand_off"][r + 1], 12])[0] + ora["cand_off"][r]) for r in range(12)]
    assert list(ora["cands"][act[8:], 2]) == [-1] * 4                # the junk pairs did not map
    assert [int(post["post"][i][5]) for i in act] == [0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 1]


def test_split_read(case):
    """CheckSplitReads (split.go:31-163): R1 = 80 bases from A followed by 70 bases from B (50 kb away), its mate next to A.
    Primary = the A part ([0,80) of the read), the B part overlaps it by 0 < 70/2 bases and scores >= 36: it becomes the split.
    One candidate => mapq = min(60, score); second_best = scoreAlignment(primary, nil) + pseudo count
    = (-5 - 70/2 + penalty) + (-10 - (150-25)/2); score = scoreAlignment(split, mate) = -5 - 80/2 + 0 + penalty."""
    A, B, pen = 200000, 250000, -4
    r1 = np.concatenate([case.c1[A:A + 80], case.c1[B + 80:B + RL]])
    ora, post = case.run([(r1, case.rev(A + 200))])
    c, off = ora["cands"], ora["cand_off"]
    assert off[1] - off[0] == 2
    rows = {int(c[i, 2]) // 10000: i for i in range(off[0], off[1])}   # by locus
    ia, ib = rows[20], rows[25]
    qa, qb_ = post["post"][ia][:2], post["post"][ib][:2]
    assert qa[0] == 0 and 78 <= qa[1] <= 84 and 76 <= qb_[0] <= 82 and qb_[1] == RL
    assert c[ia, 12] == 1 and c[ib, 12] == 0                         # the A part pairs with the mate: active
    s = post["split"][0]
    assert s[0] == ib and s[3] == 1 and s[4] == 1
    assert s[1] == min(60, c[ib, 6])
    assert s[2] == 0                                                 # B is 50 kb from the mate: not a proper pair
    assert s[5] == c[ia, 11] + 2 * pen - 20 - (RL - 25)
    assert s[6] == c[ib, 11] + c[off[1], 11] + 2 * pen
    assert c[ia, 11] == -(10 + (RL - qa[1])) - 4 * c[ia, 7]          # lap2 of the primary: one soft clip of the uncovered tail
    assert post["split"][1][0] == -1


def test_split_respects_centromere_and_needs_15_uncovered_bases(case):
    """split.go:48-50: a primary that leaves fewer than 15 bases uncovered is never split; split.go:122-133: a split candidate inside
    the centromere row of its contig gets mapq 0."""
    A, B = 200000, 250000
    r1 = np.concatenate([case.c1[A:A + 80], case.c1[B + 80:B + RL]])
    cen = (np.array([0, 240000]), np.array([0, 260000]))
    ora, post = case.run([(r1, case.rev(A + 200))], centromeres=cen)
    assert post["split"][0][0] >= 0 and post["split"][0][1] == 0
    r1 = np.concatenate([case.c1[A:A + 138], case.c1[B + 138:B + RL]])      # 12 uncovered bases: Pe - Ps = 138 > 150 - 15
    ora, post = case.run([(r1, case.rev(A + 200))])
    assert post["split"][0][0] == -1


@pytest.mark.gpu
def test_known_answers_on_gpu(built):
    c = Case(api.LIB_PATH)
    test_walk_forward_and_reverse(c)
    test_duplicates_first_pair_wins(c)
    test_split_read(c)
