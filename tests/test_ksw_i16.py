"""The 16-bit element size of ksw_align2 (ksw.c:232-334): mem_matesw takes it for mates of 250 bases and more (bwamem_pair.c:150).  CPU
tests: the oracle's restatement against the fixture the compiled reference wrote (tests/golden/ksw_i16_v1.npz, make_i16_golden.py) and, where
oracle/_ref is present, against the reference live; the device logic (host-compiled test double) against the same fixture."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import parity
import refdrv
import workloads
from arachne_amd import api, synth

GOLD = os.path.join(workloads.GOLDEN_DIR, "ksw_i16_v1.npz")
SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libarx_hostsim.so")
KSW_XBYTE, KSW_XSUBO, KSW_XSTART = 0x10000, 0x40000, 0x80000


def _cases(z):
    qo = np.concatenate([[0], np.cumsum(z["sw_qlen"])])
    to = np.concatenate([[0], np.cumsum(z["sw_tlen"])])
    for i in range(len(z["sw_qlen"])):
        yield i, z["sw_q"][qo[i]:qo[i + 1]], z["sw_t"][to[i]:to[i + 1]]


def _index(z, tag, builder):
    tmp = tempfile.mkdtemp(prefix="arx_i16_")
    prefix = os.path.join(tmp, tag + ".fa")
    synth.Genome(["c0"], [z[tag + "_contig"]], [False]).write_fasta(prefix)
    builder(prefix, prefix)
    return prefix


@pytest.fixture(scope="module")
def gold(built):
    import oradrv
    z = np.load(GOLD)
    prefix = _index(z, "mix250", api.index_build)
    o = oradrv.Oracle(prefix)
    yield z, o, prefix
    o.close()


def test_oracle_ksw_i16_against_the_reference_fixture(gold):
    z, o, _ = gold
    for i, q, t in _cases(z):
        for k, x in enumerate(z["sw_xtra"]):
            got = o.ksw_align2(q, t, int(x))
            assert (got == z["sw_out"][i, k]).all(), (i, k, got, z["sw_out"][i, k])
    assert int(z["sw_out"][:, 0, 0].max()) >= 250


def test_oracle_dispatch_follows_the_flag_not_the_length(gold):
    """ksw_align2 picks the element size from KSW_XBYTE alone (ksw.c:343-360); the two kernels agree wherever the 8-bit one cannot
    saturate, so a short query gives the same answer both ways -- and the forced 16-bit entry equals the dispatch without the flag."""
    z, o, _ = gold
    base = KSW_XSUBO | KSW_XSTART | 19
    for i, q, t in _cases(z):
        if i % 4:
            continue
        wide = o.ksw_align2(q, t, base)
        forced = o.ksw_align2_i16(q, t, base)
        assert (wide == forced).all(), i
        if len(q) < 250:
            assert (o.ksw_align2(q, t, base | KSW_XBYTE) == wide).all(), i


@pytest.mark.parametrize("tag", ["mix250", "all255"])
def test_oracle_pair_path_on_long_reads_against_the_reference_fixture(gold, tag):
    import oradrv
    z, o, prefix = gold
    oo = o if tag == "mix250" else oradrv.Oracle(_index(z, tag, api.index_build))
    oo.counters(reset=True)
    out = oo.batch(z[tag + "_reads"], z[tag + "_lens"], n_threads=2)
    assert oo.counters()["n_u8_calls"] > 40
    for k in ("reg_off", "regs", "alns", "cigars"):
        assert out[k].shape == z[tag + "_" + k].shape and (out[k] == z[tag + "_" + k]).all(), k
    if oo is not o:
        oo.close()


@pytest.mark.parametrize("tag", ["mix250", "all255"])
def test_device_logic_on_long_reads_against_the_reference_fixture(gold, tag):
    """The host-compiled double of the device functors (dev_sw.h u8_pass with eight stripes, the per-mate element size in KSwU8, the
    widest extension / CIGAR classes) on the same pairs."""
    z, _, prefix = gold
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    if tag != "mix250":
        prefix = _index(z, tag, api.index_build)
    ref = api.Reference(prefix, lib_path=SIM)
    dev = ref.batch(z[tag + "_reads"], z[tag + "_lens"]).run().fetch()
    parity.check_final(dev, {k: z[tag + "_" + k] for k in ("reg_off", "regs", "alns", "cigars")})
    ref.close()


def test_device_logic_refuses_256_bases(gold):
    z, _, prefix = gold
    ref = api.Reference(prefix, lib_path=SIM)
    with pytest.raises(api.ArachneError, match="255"):
        ref.batch(np.zeros((2, 256), np.uint8), np.array([256, 256], np.int32))
    ref.close()


@pytest.mark.skipif(not refdrv.available(), reason="oracle/_ref/libbwaref.so not built")
def test_oracle_ksw_i16_against_the_reference_live(gold):
    z, o, prefix = gold
    r = refdrv.Ref(prefix)
    rng = np.random.default_rng(77)
    for it in range(400):
        ql = int(rng.integers(1, 256))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        t = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 100)), dtype=np.uint8), q, rng.integers(0, 4, size=int(rng.integers(0, 100)), dtype=np.uint8)])
        m = rng.random(len(t)) < rng.choice([0.0, 0.03, 0.1, 0.3])
        t[m] = rng.integers(0, 5, size=int(m.sum()), dtype=np.uint8)
        if it % 7 == 0:
            t = np.delete(t, slice(len(t) // 2, len(t) // 2 + int(rng.integers(1, 12))))
        for x in (KSW_XSUBO | KSW_XSTART | 19, KSW_XSUBO | 40):
            assert (o.ksw_align2(q, t, x) == r.ksw_align2(q, t, x)).all(), (it, x)
    r.close()
