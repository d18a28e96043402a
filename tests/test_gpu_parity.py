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


@pytest.mark.parametrize("seed,no_ahead,sw_filter", [(21, False, False), (22, False, False), (22, True, False), (21, False, True), (23, False, True)])
def test_fresh_nasty_workload_matches_oracle_and_reference(built, seed, no_ahead, sw_filter, monkeypatch):
    """no_ahead: every rescue SW takes the one-at-a-time path instead of the queued-ahead launch (pipeline.h stage_rescue).
    sw_filter: rescue alignments that provably stay below min_seed_len never reach the DP (k_sw_filter_g16; off by default)."""
    import oradrv
    if no_ahead:
        monkeypatch.setenv("ARX_RESCUE_NO_AHEAD", "1")
    if sw_filter:
        monkeypatch.setenv("ARX_SW_FILTER", "1")
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


def _run_with_rfa(ref, rs, b0, b1):
    """Barcodes [b0, b1) as one device batch: final regions + placed candidates."""
    import rfadrv
    po = rs.pair_offsets()
    p0, p1 = int(po[b0]), int(po[b1])
    b = ref.batch(rs.seqs[2 * p0:2 * p1], rs.lens[2 * p0:2 * p1]).run()
    out = b.fetch()
    flags = [rfadrv.worth_running_rfa(rs.barcodes[i], int(po[i + 1] - po[i])) for i in range(b0, b1)]
    cands = b.rfa(po[b0:b1 + 1] - po[b0], flags)
    b.free()
    return out, cands


def test_results_do_not_depend_on_batch_shape_or_schedule(built):
    """Size-independent property at bench scale (one chr20-size barcode set is 1,000 x 1,000 pairs; here 24 x 1,000 on an 8 Mb
    genome): a barcode's regions, CIGARs, placements and MAPQs are the same whether it is processed in one batch of 24 barcodes,
    in two batches of 12, or a second time -- lanes take reads in whatever order the hardware schedules them (persistent-lane
    kernels, atomically claimed work lists), none of which may show in the output."""
    from arachne_amd import synth
    g = synth.make_genome(77, [6_000_000, 2_000_000])
    rs = synth.make_reads(78, g, 24, 1000)
    tmp = tempfile.mkdtemp(prefix="arx_gpu_shape_")
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    api.index_build(prefix, prefix)
    ref = api.load_reference(prefix, 0)
    whole, wc = _run_with_rfa(ref, rs, 0, 24)
    again, ac = _run_with_rfa(ref, rs, 0, 24)
    for k in ("reg_off", "regs", "alns", "cigars"):
        assert np.array_equal(whole[k], again[k]), k
    assert np.array_equal(wc["cands"], ac["cands"])
    left, lc = _run_with_rfa(ref, rs, 0, 12)
    right, rc = _run_with_rfa(ref, rs, 12, 24)
    n_left = len(left["reg_off"]) - 1
    assert np.array_equal(whole["reg_off"][:n_left + 1], left["reg_off"])
    assert np.array_equal(whole["reg_off"][n_left:] - whole["reg_off"][n_left], right["reg_off"])
    cut = int(left["reg_off"][-1])
    assert np.array_equal(whole["regs"][:cut], left["regs"]) and np.array_equal(whole["regs"][cut:], right["regs"])
    names = [n for n in wc["cands"].dtype.names if n not in ("reg", "read")]     # indices are batch-relative
    nc = len(lc["cands"])
    for n in names:
        assert np.array_equal(wc["cands"][n][:nc], lc["cands"][n]), n
        assert np.array_equal(wc["cands"][n][nc:], rc["cands"][n]), n
    act = wc["cands"][wc["cands"]["active"] == 1]
    assert len(act) == 2 * rs.n_pairs                                             # one placement per read
    ref.close()


def test_interval_pool_overflow_is_reported(env, monkeypatch):
    """Running out of the batch-wide interval pool of the seeding passes (ARX_SEED_POOL entries per read) must come back as an
    error of arx_batch_run: no silent loss of seeds, no out-of-bounds access."""
    z, ref, o = env
    monkeypatch.setenv("ARX_SEED_POOL", "2")
    with pytest.raises(api.ArachneError):
        ref.batch(z["reads"][:200], z["lens"][:200]).run()
    monkeypatch.delenv("ARX_SEED_POOL")
    dev = ref.mem_mate_sw(z["reads"][:200], z["lens"][:200])      # the context is still usable afterwards
    parity.check_final(dev, o.batch(z["reads"][:200], z["lens"][:200]))


@pytest.mark.parametrize("read_len,mixed", [(255, False), (250, False), (250, True), (249, False), (200, False), (101, False)])
def test_other_read_lengths_take_the_wide_kernel_variants(built, read_len, mixed):
    """Reads up to MAX_READ_LEN = 255: the widest extension class, wide CIGAR bands, the 16-stripe rescue SW below 250 bases and -- round 3 --
    ksw_i16's eight stripes from 250 on (ksw_align2 drops KSW_XBYTE there, bwamem_pair.c:150), also in a batch that mixes both; against the
    restatement AND the compiled reference."""
    import oradrv
    import refdrv
    import rfadrv
    g, rs, seqs, lens = workloads.long_reads(read_len, mixed=mixed)
    tmp = tempfile.mkdtemp(prefix="arx_gpu_len_")
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    api.index_build(prefix, prefix)
    ref = api.load_reference(prefix, 0)
    o = oradrv.Oracle(prefix)
    b = ref.batch(seqs, lens).run()
    o.counters(reset=True)
    ora = o.batch(seqs, lens, n_threads=8)
    assert o.counters()["n_u8_calls"] > 50                       # the rescue SW really ran
    dev = b.fetch()
    parity.check_final(dev, ora)
    if refdrv.available():
        r = refdrv.Ref(prefix)
        parity.check_final(dev, r.batch(seqs, lens, n_threads=8))
        r.close()
    po = rs.pair_offsets()
    names, offs, clens, alt, l_pac = ref.contigs()
    parity.check_rfa(b.rfa(po, [True] * 3), rfadrv.oracle_rfa(ora, lens, po, [True] * 3, l_pac, offs))
    b.free()
    ref.close()


def test_index_info_and_the_sampled_suffix_array_fallback(built, monkeypatch):
    """arx_index_info reports what arx_open built; with ARX_TEXT_INDEX=0 (what a device short of memory gets by itself: no whole suffix array, no
    inverse, every extension base by base, locate by the walk to the sample every 4th row) the results are the same."""
    import oradrv
    g, rs, seqs, lens = workloads.long_reads(150, n_bc=3, ppb=300)
    tmp = tempfile.mkdtemp(prefix="arx_gpu_info_")
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    api.index_build(prefix, prefix)
    ref = api.load_reference(prefix, 0)
    info = ref.index_info()
    assert info["symbols"] == 2 * 1_200_000 and info["text_mode"] and info["sa_rows_per_entry"] == 1 and info["kmer_k"] == 10 and info["kmer_fwd_depth"] == 10
    assert info["device_bytes"] > 2 * 5 * info["symbols"]
    a = ref.batch(seqs, lens).run().fetch()
    ref.close()
    monkeypatch.setenv("ARX_TEXT_INDEX", "0")
    ref = api.load_reference(prefix, 0)
    info = ref.index_info()
    assert not info["text_mode"] and info["sa_rows_per_entry"] == 4
    b = ref.batch(seqs, lens).run().fetch()
    ref.close()
    o = oradrv.Oracle(prefix)
    parity.check_final(a, o.batch(seqs, lens, n_threads=8))
    for k in ("reg_off", "regs", "cigars"):
        assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
    for name in a["alns"].dtype.names:
        assert (a["alns"][name] == b["alns"][name]).all(), name


def test_reads_of_256_bases_are_refused(built):
    from arachne_amd import synth
    g = synth.make_genome(5, [300_000])
    tmp = tempfile.mkdtemp(prefix="arx_gpu_len_")
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    api.index_build(prefix, prefix)
    ref = api.load_reference(prefix, 0)
    with pytest.raises(api.ArachneError, match="255"):
        ref.batch(np.zeros((2, 256), np.uint8), np.array([256, 256], np.int32))
    ref.close()


@pytest.mark.gpu
def test_wave_introsort_equals_klib_introsort(built):
    """The order klib's introsort leaves equal keys in is part of the reference's results (mem_sort_dedup_patch, mem_chain_flt).  The
    wavefront-per-item kernels reproduce the algorithm with 64 lanes (dev_regs_wave.h: w_introsort); here against the one-thread
    original on 20,000 random index arrays (2..832 entries; few, some or hardly any equal keys; random, sorted, reversed and nearly
    sorted input)."""
    assert api.selftest_wave_sort(20000, seed=20251004) == 0
    assert api.selftest_wave_sort(20000, seed=7) == 0
