"""CPU tests: the oracle restatement (oracle/arx_oracle.c) against the committed golden vectors
(tests/golden/bwa_path_v1.npz, produced from the reference's compiled C core by make_golden.py)."""
import os
import tempfile

import numpy as np
import pytest

import workloads

GOLD = os.path.join(workloads.GOLDEN_DIR, "bwa_path_v1.npz")
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


@pytest.fixture(scope="module")
def gold(built):
    import oradrv
    z = np.load(GOLD)
    tmp = tempfile.mkdtemp(prefix="arx_gold_")
    prefix = workloads.unpack_index(z, tmp)
    o = oradrv.Oracle(prefix)
    yield z, o
    o.close()


def test_fm_index_kats(gold):
    z, o = gold
    assert (o.occ4(z["kat_occ_k"]) == z["kat_occ_out"]).all()
    assert (o.sa(z["kat_sa_k"]) == z["kat_sa_out"]).all()


def test_collect_intv(gold):
    z, o = gold
    off = z["kat_intv_off"]
    for i in range(len(off) - 1):
        got = o.collect_intv(z["reads"][i])
        exp = z["kat_intv"][off[i]:off[i + 1]]
        assert got.shape == exp.shape and (got == exp).all(), i


@pytest.mark.parametrize("flt", [0, 1])
def test_chains(gold, flt):
    z, o = gold
    co = z["kat_chainf_off" if flt else "kat_chain_off"]
    so = z["kat_seedf_off" if flt else "kat_seed_off"]
    C_ = z["kat_chainf" if flt else "kat_chain"]
    S_ = z["kat_seedf" if flt else "kat_seed"]
    for i in range(len(co) - 1):
        c, s, fr = o.chains(z["reads"][i], flt)
        ec = C_[co[i]:co[i + 1]].copy()
        ec[:, 3] -= ec[0, 3] if len(ec) else 0
        assert c.shape == ec.shape and (c == ec).all(), i
        assert (s == S_[so[i]:so[i + 1]]).all(), i
        if len(c):
            assert fr == z["kat_fracrep"][i]


def test_align1_regs(gold):
    z, o = gold
    off = z["kat_reg_off"]
    for i in range(len(off) - 1):
        got = o.align1(z["reads"][i])
        exp = z["kat_reg"][off[i]:off[i + 1]]
        assert got.shape == exp.shape and (got == exp).all(), i


def _sw_cases(z):
    qo = np.concatenate([[0], np.cumsum(z["sw_qlen"])])
    to = np.concatenate([[0], np.cumsum(z["sw_tlen"])])
    for i in range(len(z["sw_qlen"])):
        yield i, z["sw_q"][qo[i]:qo[i + 1]], z["sw_t"][to[i]:to[i + 1]]


def test_ksw_extend2(gold):
    z, o = gold
    for i, q, t in _sw_cases(z):
        w, eb, zd, h0 = z["sw_ext_par"][i]
        assert (o.ksw_extend2(q, t, int(w), int(eb), int(zd), int(h0)) == z["sw_ext_out"][i]).all(), i


def test_ksw_align2_u8(gold):
    z, o = gold
    for i, q, t in _sw_cases(z):
        got = o.ksw_align2(q, t, KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 19)
        assert (got == z["sw_aln_out"][i]).all(), (i, got, z["sw_aln_out"][i])


def test_ksw_global2(gold):
    z, o = gold
    off = z["sw_glo_off"]
    for i, q, t in _sw_cases(z):
        w, sc = z["sw_glo_par"][i]
        s, cg = o.ksw_global2(q, t, int(w))
        assert s == sc, i
        assert (cg == z["sw_glo_cig"][off[i]:off[i + 1]]).all(), i


def test_pair_path(gold):
    z, o = gold
    out = o.batch(z["reads"], z["lens"], score_delta=25, n_threads=2)
    assert (out["reg_off"] == z["pair_reg_off"]).all()
    assert (out["regs"] == z["pair_regs"]).all()
    assert (out["alns"] == z["pair_alns"]).all()
    assert (out["cigars"] == z["pair_cigars"]).all()
    c = o.counters()
    assert c["ext_same_block"] > 0 and c["sa_lookups"] > 0
