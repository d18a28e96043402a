"""The shapes of BASELINE.json's configs that the generic parity tests do not reach, whole path through the C ABI
against the restatements (bit-exact: regions, CIGARs, every candidate field incl. MAPQ, post passes):

  * segmental-duplication reads: a 60-copy 5 kb family at 1 % -- region lists of 50-120 entries per read, B-tree splits in chaining,
    the quadratic loops of the rescue and placement code (what configs[2]/[3] spend their time on);
  * configs[0]: a 4.6 Mbp genome, ONE barcode of 10,000 pairs (20,000 reads in one RFA workgroup: the global-memory sort path of
    hip_block.h, > 4096 candidates);
  * the reader's limits: one barcode of 30,000 pairs (reader.go:236) and a 201-pair non-unique chunk (reader.go:266-270: no RFA);
  * exact ties: a barcode whose reads come from exact duplicate copies -- tagBestAlignments' score ties, where the reference adds
    md5-seeded jitter (aligner.go:1415-1431) and this implementation documents "first pair wins" (asserted against the restatement).

Each has a host-double variant on a reduced size (CPU suite) and a GPU variant at the stated size (-m gpu).
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import parity
import rfadrv
from arachne_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim", "libarx_hostsim.so")


def _build(g, alt=False):
    d = tempfile.mkdtemp(prefix="arx_shape_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    if alt:
        g.write_alt(fa + ".alt")
    return fa


def _run_and_check(lib_path, g, rs, do_rfa=None, post=True, stages=False, threads=8, alt=False, also_ref=False):
    import oradrv
    fa = _build(g, alt)
    api.index_build(fa, fa, lib_path=lib_path if lib_path != SIM else api.LIB_PATH)
    o = oradrv.Oracle(fa)
    ref = api.Reference(fa, lib_path=lib_path)
    try:
        po = rs.pair_offsets()
        flags = do_rfa if do_rfa is not None else [rfadrv.worth_running_rfa(rs.barcodes[b], int(po[b + 1] - po[b])) for b in range(len(po) - 1)]
        names, offs, clens, alt, l_pac = ref.contigs()
        ob = o.batch(rs.seqs, rs.lens, n_threads=threads)
        b = ref.batch(rs.seqs, rs.lens).run()
        if stages:
            heavy = np.argsort(-np.diff(ob["reg_off"]))[:40]            # the reads with the longest region lists
            parity.check_intervals(b, o, rs.seqs, rs.lens, reads=heavy)
            parity.check_chains(b, o, rs.seqs, rs.lens, reads=heavy)
            parity.check_core(b, o, rs.seqs, rs.lens, reads=heavy)
        dev = b.fetch()
        parity.check_final(dev, ob)
        if also_ref:
            import refdrv
            if refdrv.available():                                       # the reference's own C core on the same reads
                r = refdrv.Ref(fa)
                parity.check_final(dev, r.batch(rs.seqs, rs.lens, n_threads=threads))
                r.close()
        ora = rfadrv.oracle_rfa(ob, rs.lens, po, flags, l_pac, offs)
        cands = b.rfa(po, flags)
        parity.check_rfa(cands, ora)
        if post:
            parity.check_post(b.post(), rfadrv.oracle_post(o.h, ob, rs.seqs, rs.lens, po, offs, ora))
        act = cands["cands"][cands["cands"]["active"] == 1]
        assert len(act) == len(rs.lens)
        b.free()
        return dev, cands, ob
    finally:
        ref.close()
        o.close()


def _segdup_workload(seed, genome_len, copies, n_bc, ppb):
    g = synth.make_genome(seed, [genome_len, 40000], repeat_families=[(copies, 5000, 0.01), (30, 300, 0.10)], n_runs=1)
    # molecules drawn uniformly: with copies * 5 kb of the genome in the family, that share of the reads sits in it
    rs = synth.make_reads(seed + 1, g, n_bc, ppb, molecule_len=20000, molecules_per_barcode=6)
    return g, rs


def test_segdup_reads_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    g, rs = _segdup_workload(71, 1_200_000, 60, 4, 60)        # a quarter of the genome is the 60-copy family (copies may overwrite each other)
    dev, cands, ob = _run_and_check(SIM, g, rs, stages=True)
    assert np.diff(ob["reg_off"]).max() >= 25                  # the long lists are really there


def test_heavy_item_split_hostsim(built, monkeypatch):
    """The hand-over of heavy items (pairs with long region lists, reads with many seed occurrences) from the thread-per-item kernels to
    their own launches, on the host double: flagging, the lists, the skip in the main kernel (the host double runs the same serial code
    on the listed items; the wavefront versions are compared with the restatements in the GPU tests)."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    monkeypatch.setenv("ARX_SIM_RESCUE_HEAVY", "1")
    monkeypatch.setenv("ARX_SIM_CHAIN_HEAVY", "1")
    monkeypatch.setenv("ARX_SIM_DEDUP_HEAVY", "1")
    monkeypatch.setenv("ARX_DEDUP_HEAVY_MIN", "3")
    monkeypatch.setenv("ARX_RESCUE_HEAVY_MIN", "6")
    monkeypatch.setenv("ARX_CHAIN_HEAVY_MIN", "12")
    g, rs = _segdup_workload(73, 900_000, 50, 3, 50)
    _run_and_check(SIM, g, rs, stages=True)


@pytest.mark.gpu
def test_segdup_reads_gpu(built):
    g, rs = _segdup_workload(72, 6_000_000, 200, 8, 250)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs, stages=True)
    assert np.diff(ob["reg_off"]).max() >= 40


@pytest.mark.gpu
@pytest.mark.parametrize("heavy_min", ["6", "48"])
def test_long_lists_with_exact_ties_wave_rescue_gpu(built, monkeypatch, heavy_min):
    """Pairs whose region lists are long AND full of exact ties (80 identical 3 kb copies next to a 50-copy family at 0.3 %): the rescue
    replay of such pairs runs in the one-wavefront-per-pair kernel (dev_regs_wave.h), ties send it through its general pass; their
    chaining (65+ seed occurrences per read, dozens of chains of equal weight) runs in the one-wavefront-per-read kernel
    (dev_chain_wave.h).  With the thresholds lowered (6 regions, 4 occurrences) nearly every pair / read of the batch takes those
    kernels; 48 / 64 are the product settings."""
    monkeypatch.setenv("ARX_RESCUE_HEAVY_MIN", heavy_min)
    monkeypatch.setenv("ARX_CHAIN_HEAVY_MIN", "4" if heavy_min == "6" else "64")       # likewise the one-wavefront-per-read chaining kernel
    monkeypatch.setenv("ARX_DEDUP_HEAVY_MIN", "2" if heavy_min == "6" else "32")       # and the one that de-duplicates a read's regions
    g = synth.make_genome(75, [2_000_000, 30000], repeat_families=[(80, 3000, 0.0), (50, 2000, 0.003)], n_runs=1)
    rs = synth.make_reads(76, g, 6, 150, molecule_len=10000, molecules_per_barcode=5)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs, stages=True)
    assert np.diff(ob["reg_off"]).max() >= 40


def _deletion_reads(seed, n_pairs):
    """Pairs whose first read skips 4-190 bp of the reference in its middle: two collinear regions from two chains, the pairs of regions
    mem_sort_dedup_patch hands to mem_patch_reg (bwamem.c:406-435, 463-470).  Its tests on the gap ratio let only the short gaps through
    to the alignment; the long ones stay two regions."""
    g = synth.make_genome(seed, [600_000], repeat_families=[(8, 3000, 0.02)], n_runs=0)
    rs = synth.make_reads(seed + 1, g, 2, n_pairs, molecule_len=20000, molecules_per_barcode=4, indel_rate=0.0)
    rng = np.random.default_rng(seed + 2)
    ref = g.seqs[0]
    for i in range(0, rs.n_pairs, 2):                                # every second pair: R1 rebuilt with a deletion, R2 a proper mate
        p = int(rng.integers(1000, len(ref) - 2000))
        d = int(rng.integers(110, 191)) if rng.random() < 0.5 else int(rng.integers(4, 12))
        cut = int(rng.integers(60, 91))
        fwd = np.concatenate([ref[p:p + cut], ref[p + cut + d:p + d + 150]])
        if rng.random() < 0.5:
            r1, r2 = fwd, synth._revcomp(ref[p + d + 250:p + d + 400])
        else:                                                        # the pair on the other strand
            r1, r2 = synth._revcomp(fwd), ref[p - 400:p - 250].copy()
        rs.seqs[2 * i], rs.seqs[2 * i + 1] = r1, r2
    return g, rs


def test_deletion_reads_collinear_regions_hostsim(built, monkeypatch):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    monkeypatch.setenv("ARX_SIM_DEDUP_HEAVY", "1")
    monkeypatch.setenv("ARX_DEDUP_HEAVY_MIN", "2")
    g, rs = _deletion_reads(83, 60)
    dev, cands, ob = _run_and_check(SIM, g, rs)
    assert int((np.diff(dev["reg_off"]) >= 2).sum()) >= 10             # reads that kept both regions


@pytest.mark.gpu
@pytest.mark.parametrize("heavy_min", ["2", "32"])
def test_deletion_reads_collinear_regions_gpu(built, monkeypatch, heavy_min):
    """The same through the GPU library; with the threshold at 2 every read with two regions takes the one-wavefront-per-read
    de-duplication kernel, whose probe must send the mergeable ones to the one-thread pass (dev_regs_wave.h: w_sort_dedup)."""
    monkeypatch.setenv("ARX_DEDUP_HEAVY_MIN", heavy_min)
    g, rs = _deletion_reads(84, 300)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs)
    assert int((np.diff(dev["reg_off"]) >= 2).sum()) >= 50


def _one_barcode(seed, genome_len, n_pairs):
    g = synth.make_genome(seed, [genome_len])
    rs = synth.make_reads(seed + 1, g, 1, n_pairs, molecules_per_barcode=12)
    return g, rs


def test_one_large_barcode_hostsim(built):
    """configs[0]'s shape at a tenth of its size: one barcode, 1,000 pairs (> the 1024 lanes' worth of reads, several molecules)."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    g, rs = _one_barcode(81, 400_000, 1000)
    _run_and_check(SIM, g, rs)


@pytest.mark.gpu
def test_config0_ecoli_size_one_barcode_of_10000_pairs_gpu(built):
    g, rs = _one_barcode(20250905 + 1, 4_641_652, 10_000)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs)
    assert len(cands["cands"]) > 4096                          # past the in-LDS sort of the RFA workgroup


@pytest.mark.gpu
def test_reader_limits_30000_pair_barcode_and_201_pair_chunk_gpu(built):
    """The largest set ReadBarcodeSet returns (30,000 pairs) next to a 201-pair chunk of a continuing barcode (flagged non-unique:
    worthRunningRFA false, the single-read fallback of aligner.go:469-477)."""
    g = synth.make_genome(91, [3_000_000])
    big = synth.make_reads(92, g, 1, 30_000, molecules_per_barcode=40)
    small = synth.make_reads(93, g, 1, 201, molecules_per_barcode=3)
    rs = synth.ReadSet(np.concatenate([big.seqs, small.seqs]), np.concatenate([big.lens, small.lens]),
                       np.concatenate([big.barcode_id, small.barcode_id + 1]), big.barcodes + ["B01C01A01D01-1"],
                       np.concatenate([big.valid, small.valid]), np.concatenate([big.truth_contig, small.truth_contig]),
                       np.concatenate([big.truth_pos, small.truth_pos]))
    flags = [rfadrv.worth_running_rfa(rs.barcodes[0], 30_000, unique=True), rfadrv.worth_running_rfa(rs.barcodes[1], 201, unique=False)]
    assert flags == [True, False]
    _run_and_check(api.LIB_PATH, g, rs, do_rfa=flags, threads=32)


def _tie_workload(seed, n_pairs):
    """Reads from EXACT duplicate copies (divergence 0): every candidate pair of a read ties in score with its twins."""
    g = synth.make_genome(seed, [500_000], repeat_families=[(6, 4000, 0.0), (4, 1500, 0.0)], n_runs=0)
    rs = synth.make_reads(seed + 1, g, 2, n_pairs, molecule_len=8000, molecules_per_barcode=4, sub_rate=0.001, indel_rate=0.0)
    return g, rs


def test_exact_ties_first_pair_wins_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    g, rs = _tie_workload(61, 120)
    dev, cands, ob = _run_and_check(SIM, g, rs)
    _assert_ties_present_and_first_wins(cands)


@pytest.mark.gpu
def test_exact_ties_first_pair_wins_gpu(built):
    g, rs = _tie_workload(62, 400)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs)
    _assert_ties_present_and_first_wins(cands)


def _assert_ties_present_and_first_wins(cands):
    """Among the candidates of a read with the same best lap2 the documented rule (tagBestAlignments restated with `>`: the first
    pair in candidate order wins) shows as: whenever the bwa-pick stage had an exact tie, the chosen candidate is not preceded by an
    equal one that would have formed the same pair score.  The field-by-field equality with the restatement is asserted by the caller;
    here: the workload really produces tied candidates (otherwise the test would be vacuous)."""
    c, off = cands["cands"], cands["cand_off"]
    tied_reads = 0
    for r in range(len(off) - 1):
        rows = c[off[r]:off[r + 1]]
        if len(rows) >= 2:
            best = rows["lap2"].max()
            if (rows["lap2"] == best).sum() >= 2 and (rows["score"] == rows["score"].max()).sum() >= 2:
                tied_reads += 1
    assert tied_reads >= 10, tied_reads


# ---- BASELINE.json configs[3]: ALT/decoy contigs + repeat-enriched stLFR-like set; configs[4]: 30 % VX:i:0 dash-less small groups
def _config3_workload(seed, primary_lens, n_alt, n_bc, ppb, n_segdup):
    """bench.py's alt_repeat recipe scaled down: the three families of SURVEY s8d + low-copy segmental duplications, .alt-flagged ALT
    contigs (diverged copies of primary slices: bwamem.c:351, 1078-1082), unflagged decoys; half of the molecules drawn from the
    planted copies, stLFR-like barcodes of few pairs."""
    total = sum(primary_lens)
    scale = total / 3.1e9
    fams = [(max(8, int(1e4 * scale * 4)), 300, 0.12, 0.1), (max(6, int(1e3 * scale * 4)), 6000, 0.05, 0.15), (max(12, int(200 * scale * 8)), 30000, 0.01, 0.04)] + \
        synth.segdup_families(n_segdup, 0.71)
    g = synth.make_genome(seed, list(primary_lens), repeat_families=fams, alt_spec=(n_alt, 20_000, 120_000, 0.01), decoy_spec=(3, 8000))
    rs = synth.make_reads(seed + 1, g, n_bc, ppb, molecules_per_barcode=2, repeat_bias=0.5, barcode_style="stlfr")
    return g, rs


def _assert_config3_shape(g, rs, dev, ob, mean_min=3.0, max_min=20):
    assert sum(g.alt) >= 2                         # ALT contigs are in the index ...
    assert int((dev["regs"]["is_alt"] != 0).sum()) > 0, "no region landed on an ALT contig"
    nreg = np.diff(ob["reg_off"])
    assert nreg.mean() >= mean_min and nreg.max() >= max_min, (nreg.mean(), nreg.max())   # ... and the lists are long


def test_config3_alt_repeat_stlfr_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    g, rs = _config3_workload(20250905 + 4, [900_000, 500_000], 3, 8, 14, 4)
    dev, cands, ob = _run_and_check(SIM, g, rs, alt=True)
    _assert_config3_shape(g, rs, dev, ob, 2.0, 10)


@pytest.mark.gpu
def test_config3_alt_repeat_stlfr_50Mbp_20k_pairs_gpu(built):
    """configs[3] at reduced size: 50 Mbp of primary contigs + 12 ALT contigs + decoys, 20,000 pairs in 666 stLFR-like barcodes, against the
    restatement AND the compiled reference (regions .. CIGARs), RFA / MAPQ / post passes against the Go-half restatement."""
    g, rs = _config3_workload(20250905 + 4, [30_000_000, 15_000_000, 5_000_000], 12, 666, 30, 20)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs, alt=True, threads=32, also_ref=True)
    _assert_config3_shape(g, rs, dev, ob, 2.5, 20)


def _config4_workload(seed, genome_len, n_bc, ppb):
    g = synth.make_genome(seed, [genome_len - 200_000, 150_000, 50_000] if genome_len > 1_000_000 else [genome_len])
    rs = synth.make_reads(seed + 1, g, n_bc, ppb, invalid_frac=0.3)
    return g, rs


def _assert_config4_shape(rs, cands):
    po = rs.pair_offsets()
    sizes = np.diff(po)
    flags = np.array([rfadrv.worth_running_rfa(rs.barcodes[b], int(sizes[b])) for b in range(len(sizes))])
    small = ~flags
    assert small.sum() >= 10 and sizes[small].max() <= 4 and all("-" not in rs.barcodes[b] for b in np.flatnonzero(small))
    share = sizes[small].sum() / rs.n_pairs
    assert 0.2 < share < 0.4, share
    # the fallback leaves the candidates un-placed by RFA: no molecule, and exactly one active candidate per read all the same
    c = cands["cands"]
    for b in np.flatnonzero(small)[:50]:
        rows = c[cands["cand_off"][2 * po[b]]:cands["cand_off"][2 * po[b + 1]]]
        assert (rows["molecule_id"] == -1).all() and int(rows["active"].sum()) == 2 * int(sizes[b])


def test_config4_vx0_small_groups_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    g, rs = _config4_workload(20250905 + 5, 600_000, 3, 110)
    dev, cands, ob = _run_and_check(SIM, g, rs)
    _assert_config4_shape(rs, cands)


@pytest.mark.gpu
def test_config4_vx0_small_groups_20k_pairs_gpu(built):
    """configs[4] at reduced size: 20,000 pairs on a 20 Mbp genome, 30 % of them in ~2,400 dash-less groups of 1-4 pairs filed among 20 RFA
    barcodes; every field against the restatements and the compiled reference."""
    g, rs = _config4_workload(20250905 + 5, 20_000_000, 20, 1000)
    dev, cands, ob = _run_and_check(api.LIB_PATH, g, rs, threads=32, also_ref=True)
    _assert_config4_shape(rs, cands)


def _fullsize(workload, tmp):
    """tools/gpu_fullsize_check.py on a bench workload at its full step size: two batch shapes give identical digests of every
    per-read output; then a slice of the same reads against the oracle AND the compiled reference, with RFA against the restatement."""
    import importlib.util
    import sys
    import oradrv
    import refdrv
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    spec = importlib.util.spec_from_file_location("gpu_fullsize_check", os.path.join(root, "tools", "gpu_fullsize_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ok, digests, stats, (ref, rs, prefix) = mod.fullsize_check(workload, cache=str(tmp))
    try:
        assert ok, (digests, stats)
        assert stats["reads"] == 2 * rs.n_pairs and stats["active"] == stats["reads"]
        po = rs.pair_offsets()
        nb = int(np.searchsorted(po, 2000, side="right")) - 1
        nb = max(nb, 1)
        n = int(po[nb])
        seqs, lens = rs.seqs[:2 * n], rs.lens[:2 * n]
        o = oradrv.Oracle(prefix)
        ob = o.batch(seqs, lens, n_threads=16)
        b = ref.batch(seqs, lens).run()
        parity.check_final(b.fetch(), ob)
        if refdrv.available():
            r = refdrv.Ref(prefix)
            parity.check_final(b.fetch(), r.batch(seqs, lens, n_threads=16))
            r.close()
        names, offs, clens, alt, l_pac = ref.contigs()
        flags = [rfadrv.worth_running_rfa(rs.barcodes[i], int(po[i + 1] - po[i])) for i in range(nb)]
        parity.check_rfa(b.rfa(po[:nb + 1], flags), rfadrv.oracle_rfa(ob, lens, po[:nb + 1], flags, l_pac, offs))
        b.free()
        o.close()
    finally:
        ref.close()


@pytest.mark.gpu
def test_config1_chr20_size_1M_pairs_full_size_gpu(built, tmp_path_factory):
    """BASELINE.json configs[1] at full size: 64,444,167 bp, 1,000 barcodes x 1,000 pairs."""
    _fullsize("chr20", tmp_path_factory.mktemp("chr20"))


@pytest.mark.gpu
def test_config2_grch38_size_slice_full_step_gpu(built, tmp_path_factory):
    """BASELINE.json configs[2] on one GPU: the GRCh38-size index (built in HBM by arx_index_build) and one full bench step
    (13,000 TELLseq-like barcodes x 77 pairs); the slice check loads the 5.4 GB index into the reference's C core as well."""
    _fullsize("grch38", tmp_path_factory.mktemp("grch38"))


@pytest.mark.gpu
def test_config3_alt_repeat_full_step_gpu(built, tmp_path_factory):
    """BASELINE.json configs[3] at full size: GRCh38-size genome + 220 ALT contigs + 400 decoys, one full bench step of 33,000
    stLFR-like barcodes x 30 pairs, half of the molecules in planted repeats: two batch shapes give identical digests of every
    per-read output, then a slice against the restatements and the compiled reference."""
    _fullsize("alt_repeat", tmp_path_factory.mktemp("alt_repeat"))


@pytest.mark.gpu
def test_config4_vxmix_full_step_gpu(built, tmp_path_factory):
    """BASELINE.json configs[4] at full size: 1 M pairs, 30 % of them in ~120,000 dash-less groups of 1-4 pairs among 1,000 RFA barcodes."""
    _fullsize("vxmix", tmp_path_factory.mktemp("vxmix"))
