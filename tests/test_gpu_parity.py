"""GPU parity tests (run with -m gpu on an MI355X): libarachne_amd.so through its C ABI vs the oracle, the committed
golden vectors and -- where the prebuilt oracle/_ref/libbwaref.so travelled with the snapshot -- the reference's own
compiled C core.  Bit-exact: regions, order, pos, strand, NM, CIGAR."""
import os
import tempfile

import numpy as np
import pytest

import parity
import refdrv
import workloads
from arachne_amd import api, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(workloads.GOLDEN_DIR, "bwa_path_v1.npz")


@pytest.fixture(scope="module")
def env(built):
    import oradrv
    z = np.load(GOLD)
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_gpu_"))
    ref = api.load_reference(prefix, 0)
    assert ref.backend == "hip:gfx950"
    o = oradrv.Oracle(prefix)
    yield z, ref, o
    ref.close()
    o.close()


def test_stage_outputs_match_oracle(env):
    z, ref, o = env
    seqs, lens = z["reads"][:600], z["lens"][:600]
    b = ref.batch(seqs, lens).run()
    parity.check_intervals(b, o, seqs, lens)
    parity.check_chains(b, o, seqs, lens)
    parity.check_core(b, o, seqs, lens)
    b.free()


def test_pair_path_matches_golden(env):
    z, ref, o = env
    dev = ref.mem_mate_sw(z["reads"], z["lens"])
    gold = dict(reg_off=z["pair_reg_off"], regs=z["pair_regs"], alns=z["pair_alns"], cigars=z["pair_cigars"])
    parity.check_final(dev, gold)


def test_ragged_and_degenerate_reads(env):
    z, ref, o = env
    rows = [z["reads"][i] for i in range(64)]
    rows[1] = rows[1][:0]
    rows[4] = rows[4][:18]
    rows[6] = rows[6][:19]
    rows[9] = np.full(150, 4, dtype=np.uint8)
    rows[12] = rows[12][:77]
    rows[15] = np.zeros(150, dtype=np.uint8)
    lens = np.array([len(r) for r in rows], dtype=np.int32)
    flat = np.concatenate(rows)
    parity.check_final(ref.mem_mate_sw(flat, lens), o.batch(flat, lens))


@pytest.mark.parametrize("seed,no_ahead", [(21, False), (22, False), (22, True)])
def test_fresh_nasty_workload_matches_oracle_and_reference(built, seed, no_ahead, monkeypatch):
    """no_ahead: every rescue SW takes the one-at-a-time path instead of the queued-ahead launch (pipeline.h stage_rescue)."""
    import oradrv
    if no_ahead:
        monkeypatch.setenv("ARX_RESCUE_NO_AHEAD", "1")
    g = workloads.nasty_genome(seed, contig_lens=(200000, 120000, 50000), alt_contigs=2)
    rs = workloads.nasty_reads(seed, g, n_barcodes=8, pairs_per_barcode=500)
    tmp = tempfile.mkdtemp(prefix="arx_gpu_nasty_")
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    g.write_alt(prefix + ".alt")
    if not refdrv.available():
        pytest.skip("oracle/_ref/libbwaref.so did not travel; index builder of the reference unavailable")
    r = refdrv.Ref()
    r.index_build(prefix, prefix)
    r.open(prefix)
    ref = api.load_reference(prefix, 0)
    dev = ref.mem_mate_sw(rs.seqs, rs.lens)
    parity.check_final(dev, r.batch(rs.seqs, rs.lens, n_threads=8))
    parity.check_final(dev, oradrv.Oracle(prefix).batch(rs.seqs, rs.lens, n_threads=8))
    assert len(dev["regs"]) > 5000
    ref.close()
