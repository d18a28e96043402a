"""CPU tests of the device-side logic through the host-compiled test double (tests/hostsim): the same functors and the
same Pipeline stage sequence that libarachne_amd.so launches as HIP kernels, run as plain loops.  Checked stage by
stage against the oracle and the golden vectors."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import parity
import workloads
from arachne_amd import api

SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libarx_hostsim.so")
GOLD = os.path.join(workloads.GOLDEN_DIR, "bwa_path_v1.npz")


@pytest.fixture(scope="module")
def env(built):
    import oradrv
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    z = np.load(GOLD)
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_sim_"))
    ref = api.Reference(prefix, lib_path=SIM)
    o = oradrv.Oracle(prefix)
    yield z, ref, o
    ref.close()
    o.close()


def test_stages_against_oracle(env):
    z, ref, o = env
    seqs, lens = z["reads"][:400], z["lens"][:400]
    b = ref.batch(seqs, lens).run()
    parity.check_intervals(b, o, seqs, lens)
    parity.check_chains(b, o, seqs, lens)
    parity.check_core(b, o, seqs, lens)
    b.free()


def test_pair_path_against_golden(env):
    z, ref, o = env
    dev = ref.mem_mate_sw(z["reads"], z["lens"])
    gold = dict(reg_off=z["pair_reg_off"], regs=z["pair_regs"], alns=z["pair_alns"], cigars=z["pair_cigars"])
    parity.check_final(dev, gold)


def test_rescue_single_sw_path(env, monkeypatch):
    """Mate rescue normally computes every SW of a loop in one launch and replays the loop; a SW that could not be queued
    ahead takes the one-at-a-time path.  ARX_RESCUE_NO_AHEAD=1 sends every SW down that path."""
    z, ref, o = env
    monkeypatch.setenv("ARX_RESCUE_NO_AHEAD", "1")
    dev = ref.mem_mate_sw(z["reads"], z["lens"])
    gold = dict(reg_off=z["pair_reg_off"], regs=z["pair_regs"], alns=z["pair_alns"], cigars=z["pair_cigars"])
    parity.check_final(dev, gold)


@pytest.mark.parametrize("mode", ["check", "off"])
def test_rescue_incremental_dedup(env, monkeypatch, mode):
    """Inserting a rescued region into a list that is already a fixed point of mem_sort_dedup_patch takes a one-scan path
    (dev_regs.h dedup_insert).  check: the host double runs the general path next to it on a copy and aborts on any
    difference; off: the general path alone must give the same final result."""
    z, ref, o = env
    monkeypatch.setenv("ARX_RESCUE_CHECK" if mode == "check" else "ARX_RESCUE_FAST", "1" if mode == "check" else "0")
    dev = ref.mem_mate_sw(z["reads"], z["lens"])
    gold = dict(reg_off=z["pair_reg_off"], regs=z["pair_regs"], alns=z["pair_alns"], cigars=z["pair_cigars"])
    parity.check_final(dev, gold)


def test_ragged_and_degenerate_reads(env):
    z, ref, o = env
    rows = [z["reads"][i] for i in range(40)]
    rows[1] = rows[1][:0]                                # empty mate
    rows[4] = rows[4][:18]                               # shorter than a seed
    rows[6] = rows[6][:19]
    rows[9] = np.full(150, 4, dtype=np.uint8)            # all N
    rows[12] = rows[12][:77]
    rows[15] = np.zeros(150, dtype=np.uint8)             # homopolymer
    lens = np.array([len(r) for r in rows], dtype=np.int32)
    flat = np.concatenate(rows)
    dev = ref.mem_mate_sw(flat, lens)
    parity.check_final(dev, o.batch(flat, lens))


def test_argument_errors(env):
    z, ref, o = env
    with pytest.raises(api.ArachneError):
        ref.batch(z["reads"][:3], z["lens"][:3])          # odd number of reads
    with pytest.raises(api.ArachneError):
        ref.batch(np.zeros(600, dtype=np.uint8), np.array([300, 300], dtype=np.int32))  # longer than the u8 SW allows
    with pytest.raises(api.ArachneError):
        api.Reference("/nonexistent/prefix", lib_path=SIM)
    b = ref.batch(z["reads"][:40], z["lens"][:40])
    with pytest.raises(api.ArachneError):
        b.fetch()                                         # results before arx_batch_run
    with pytest.raises(api.ArachneError):
        b.rfa([0, 20], [True])                            # placement before the alignment stage
    b.run()
    with pytest.raises(api.ArachneError):
        b.post()                                          # the post-placement passes before the placement
    with pytest.raises(api.ArachneError):
        b.rfa([0, 10], [True])                            # barcode offsets must cover the batch
    with pytest.raises(api.ArachneError):
        b.rfa([5, 20], [True])
    assert len(b.rfa([0, 20], [False])["cands"]) >= 40    # still usable afterwards
    p1 = b.post()
    assert len(p1["split"]) == 40 and (p1["post"]["qe"] >= p1["post"]["qb"]).all()
    b.run()                                               # running the batch again invalidates placement and post results
    with pytest.raises(api.ArachneError):
        b.post()
    b.rfa([0, 20], [False])
    p2 = b.post()                                         # ... and the same calls give the same answer again
    assert p1["post"].tobytes() == p2["post"].tobytes() and p1["split"].tobytes() == p2["split"].tobytes()
    b.free()
    with pytest.raises(api.ArachneError):
        ref.batch(np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.int32))   # an empty batch is refused, not crashed on


def test_interval_pool_overflow_is_reported(env, monkeypatch):
    """The seeding passes keep their interval lists in a batch-wide pool sized per read (ARX_SEED_POOL entries); running out of it
    must surface as an error of arx_batch_run, never as silently missing seeds."""
    z, ref, o = env
    monkeypatch.setenv("ARX_SEED_POOL", "2")
    with pytest.raises(api.ArachneError):
        ref.batch(z["reads"][:200], z["lens"][:200]).run()


def test_close_frees_batches_left_alive(built):
    """arx_close with batches still alive frees them (their handles die with the context); the binding must not free them again."""
    z = np.load(GOLD)
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_sim_close_"))
    for _ in range(2):
        ref = api.Reference(prefix, lib_path=SIM)
        b = ref.batch(z["reads"][:60], z["lens"][:60]).run()
        ref.close()
        assert b.h is None
        del b


def test_dedup_insert_equals_the_general_pass_on_colliding_lists():
    """dedup_insert() (dev_regs.h: what a rescued region costs to merge into a clean list) against mem_sort_dedup_patch's restatement on
    random lists built so that redundant neighbours, stoppers on either side and survivors all occur (also lists beyond 256 regions:
    the serial variant)."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    lib = C.CDLL(SIM)
    lib.arx_test_dedup_insert.restype = C.c_long
    nf, ng, nb = C.c_long(), C.c_long(), C.c_long()
    cases = lib.arx_test_dedup_insert(5, 20000, C.byref(nf), C.byref(ng), C.byref(nb))
    assert cases > 10000, cases                      # negative: index of the first mismatch
    assert nf.value > 5000 and ng.value > 200 and nb.value > 200, (nf.value, ng.value, nb.value)


def test_gapfree_counts_wordwise_equals_pairwise():
    """gapfree_counts (dev_sw.h: the NM / score walk of bwa_gen_cigar2's gap-free shortcut, bwa.c:141-149,169-199, four pairs per step)
    against the pair-by-pair walk: both strands, every byte phase, strand ends, ambiguous read bases, unaligned reads."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    lib = C.CDLL(SIM)
    lib.arx_test_gapfree.restype = C.c_long
    lib.arx_test_gapfree.argtypes = [C.c_uint, C.c_int]
    for seed in (1, 2, 3):
        assert lib.arx_test_gapfree(seed, 20000) == 8 * 20000       # negative: index of the first mismatch


def test_text_mode_word_compare_equals_base_by_base():
    """Text mode of the first-pass forward extensions (dev_fm.h): the 64-bases-at-a-time comparison of a 4-bit-coded read row with the packed
    reference (what the wavefront kernels run) against the base-by-base form and the text itself -- both strands, chunk ends at the strand
    boundary and at the ends of the text, ambiguous bases, reads that end inside a chunk."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    lib = C.CDLL(SIM)
    lib.arx_test_text_match.restype = C.c_long
    for seed in (1, 2):
        assert lib.arx_test_text_match(seed, 15000) == 8 * 15000       # negative: index of the first mismatch


def test_detached_results_survive_the_next_run(env):
    """arx_batch_detach / arx_batch_fetch_detached: the results of one super-batch are taken home after the handle has been reset and run
    with the next one (what a worker with two host threads per handle does)."""
    z, ref, o = env
    import rfadrv
    seqs, lens = z["reads"], z["lens"]
    n = (len(lens) // 4) * 2
    po = np.array([0, n // 2], dtype=np.int64)
    b = ref.batch(seqs[:n], lens[:n]).run()
    first = b.fetch()
    cands = b.rfa(po, [True])
    sizes = b.detach()
    assert sizes["n_reads"] == n and sizes["n_regs"] == len(first["regs"]) and sizes["n_cands"] == len(cands["cands"])
    b.reset(seqs[n:2 * n], lens[n:2 * n]).run()                     # the handle moves on
    second = b.fetch()
    buf = {}
    b.fetch_detached_into(buf, sizes)
    assert (buf["reg_off"][:n + 1] == first["reg_off"]).all() and buf["regs"][:sizes["n_regs"]].tobytes() == first["regs"].tobytes()
    assert buf["alns"][:sizes["n_regs"]].tobytes() == first["alns"].tobytes() and (buf["cigars"][:sizes["n_cigar"]] == first["cigars"]).all()
    assert (buf["cand_off"][:n + 1] == cands["cand_off"]).all() and buf["cands"][:sizes["n_cands"]].tobytes() == cands["cands"].tobytes()
    assert second["regs"].tobytes() != first["regs"].tobytes()
    b.free()


@pytest.mark.parametrize("text", ["1", "0"])
def test_backward_sweeps_entry_by_entry(monkeypatch, text):
    """What the entry-parallel backward kernel (k_seed_bwd_e) rests on: every interval of a forward list can be extended to the left on its
    own, and the SMEMs are those whose start lies left of the last one's, in list order -- on the nasty workload (repeats, ties, ambiguous
    bases, ALT contigs), with and without text mode, interval for interval against the restatement."""
    import oradrv
    monkeypatch.setenv("ARX_SEED_BWD_ENTRY", "1")
    monkeypatch.setenv("ARX_TEXT_INDEX", text)
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    z = np.load(GOLD)
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_sim_e_"))
    ref = api.Reference(prefix, lib_path=SIM)
    o = oradrv.Oracle(prefix)
    seqs, lens = z["reads"], z["lens"]
    b = ref.batch(seqs, lens).run()
    parity.check_intervals(b, o, seqs, lens)
    parity.check_final(b.fetch(), o.batch(seqs, lens, n_threads=4))
    b.free()
    ref.close()
    o.close()
    g = workloads.nasty_genome(31, contig_lens=(90000, 40000, 15000), alt_contigs=2)
    rs = workloads.nasty_reads(31, g, n_barcodes=3, pairs_per_barcode=250)
    d = tempfile.mkdtemp(prefix="arx_sim_e2_"); fa = os.path.join(d, "g.fa")
    g.write_fasta(fa); g.write_alt(fa + ".alt")
    api.index_build(fa, fa, lib_path=SIM)
    ref = api.Reference(fa, lib_path=SIM)
    o = oradrv.Oracle(fa)
    b = ref.batch(rs.seqs, rs.lens).run()
    parity.check_intervals(b, o, rs.seqs, rs.lens)
    b.free(); ref.close(); o.close()


def test_extension_with_32_bit_child_sizes(env):
    """k_seed_bwd_g<true>'s arithmetic (dev_fm.h ext_finish<true>: the four children's sizes as 32-bit differences, the 40-bit count for one symbol)
    against the general form and extend1 on random walks over the golden index."""
    import ctypes as C
    z, ref, o = env
    lib = ref.lib
    lib.arx_test_ext_fit32.restype = C.c_long
    lib.arx_test_ext_fit32.argtypes = [C.c_void_p, C.c_uint, C.c_int]
    n = lib.arx_test_ext_fit32(ref.h, 11, 4000)
    assert n > 30000, n


def test_index_info(env):
    z, ref, o = env
    info = ref.index_info()
    assert info["symbols"] > 0 and info["text_mode"] and info["sa_rows_per_entry"] == 1 and info["kmer_k"] >= 4 and info["kmer_fwd_depth"] == min(info["kmer_k"], 14)


def test_text_mode_off_gives_the_same_seeds(env, monkeypatch):
    """ARX_TEXT_INDEX=0 (no whole suffix array / inverse: every extension base by base, locate by the sampled walk) against the default."""
    z, ref, o = env
    seqs, lens = z["reads"][:300], z["lens"][:300]
    a = ref.batch(seqs, lens).run().fetch()
    monkeypatch.setenv("ARX_TEXT_INDEX", "0")
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_sim_"))
    ref2 = api.Reference(prefix, lib_path=SIM)
    assert not ref2.index_info()["text_mode"] and ref2.index_info()["sa_rows_per_entry"] == 4
    b = ref2.batch(seqs, lens).run().fetch()
    ref2.close()
    for k in ("reg_off", "regs", "cigars"):
        assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
    for name in a["alns"].dtype.names:
        assert (a["alns"][name] == b["alns"][name]).all(), name


def test_batch_reset_reuses_the_handle_and_matches_a_fresh_batch():
    """arx_batch_reset: new reads into an existing handle (larger, then smaller than the first set) give what a fresh batch gives."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    z = np.load(os.path.join(workloads.GOLDEN_DIR, "bwa_path_v1.npz"))
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_reset_"))
    ref = api.Reference(prefix, lib_path=SIM)
    reads, lens = z["reads"], z["lens"]
    b = ref.batch(reads[:100], lens[:100]).run()
    first = b.fetch()
    for lo, hi in ((100, 500), (40, 60), (0, 100)):
        fresh = ref.mem_mate_sw(reads[lo:hi], lens[lo:hi])
        again = b.reset(reads[lo:hi], lens[lo:hi]).run().fetch()
        for k in ("reg_off", "regs", "alns", "cigars"):
            assert np.array_equal(fresh[k], again[k]), (lo, hi, k)
        po = [0, (hi - lo) // 4, (hi - lo) // 2]
        c1 = b.rfa(po, [True, True])
        b2 = ref.batch(reads[lo:hi], lens[lo:hi]).run()
        c2 = b2.rfa(po, [True, True])
        assert np.array_equal(c1["cands"], c2["cands"]) and np.array_equal(c1["cand_off"], c2["cand_off"])
        buf = {}
        b.fetch_into(buf)
        assert np.array_equal(buf["regs"][:len(again["regs"])], again["regs"]) and np.array_equal(buf["cands"][:len(c1["cands"])], c1["cands"])
        b2.free()
    assert np.array_equal(first["regs"], b.reset(reads[:100], lens[:100]).run().fetch()["regs"])
    with pytest.raises(api.ArachneError):
        b.reset(reads[:3], lens[:3])          # odd number of reads
    b.free()
    ref.close()
