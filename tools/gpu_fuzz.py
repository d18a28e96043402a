#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (run under gpurun): fresh seeded workloads of both generators, whole path + RFA against the
CPU restatement (and the compiled reference when it travelled).  Not part of the test suite: a wider net cast once in a while."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from arachne_amd import api, synth
import oradrv, parity, refdrv, rfadrv, workloads

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
base = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "mixed"   # "repeats": every seed a high-copy family (long region lists, ties): the wavefront-per-item kernels
for seed in range(base, base + n_seeds):
    t = time.time()
    if mode == "repeats":
        r = np.random.default_rng(seed)
        fams = [(int(r.integers(30, 220)), int(r.integers(800, 6000)), float(r.choice([0.0, 0.002, 0.01, 0.03]))), (int(r.integers(10, 60)), int(r.integers(200, 1500)), float(r.choice([0.0, 0.05])))]
        g = synth.make_genome(seed, [int(r.integers(500_000, 2_500_000)), 50000], repeat_families=fams, n_runs=int(r.integers(0, 3)))
        rs = synth.make_reads(seed + 1, g, 5, 240, molecule_len=int(r.integers(4000, 30000)), molecules_per_barcode=int(r.integers(3, 8)))
    elif mode == "long":   # read lengths around the switch of ksw_align2's element size (250 bases), ragged, a third of the reads damaged
        r = np.random.default_rng(seed)
        L = int(r.choice([255, 252, 250, 249, 230]))
        g = synth.make_genome(seed, [int(r.integers(300_000, 1_500_000)), 60000], repeat_families=[(int(r.integers(5, 40)), int(r.integers(300, 2000)), 0.01)])
        rs = synth.make_reads(seed + 1, g, 4, 220, read_len=L, sub_rate=0.01)
        for i in r.choice(rs.seqs.shape[0], size=rs.seqs.shape[0] // 3, replace=False):
            m = r.random(L) < (0.1, 0.2, 0.35)[i % 3]
            rs.seqs[i, m] = r.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
        if seed % 2:
            rs.lens[r.choice(len(rs.lens), size=len(rs.lens) // 4, replace=False)] = r.integers(30, L + 1, size=len(rs.lens) // 4)
    elif seed % 3 == 0:
        g = synth.make_genome(seed, [1500000, 400000]); rs = synth.make_reads(seed + 1, g, 6, 500)
    else:
        g = workloads.nasty_genome(seed, contig_lens=(180000 + 1000 * (seed % 7), 90000, 40000), alt_contigs=seed % 3)
        rs = workloads.nasty_reads(seed, g, n_barcodes=6, pairs_per_barcode=400)
    d = tempfile.mkdtemp(prefix="arx_fuzz_"); fa = os.path.join(d, "g.fa")
    g.write_fasta(fa); g.write_alt(fa + ".alt")
    api.index_build(fa, fa)
    po = rs.pair_offsets()
    rng = np.random.default_rng(7000 + seed)   # PCR duplicates and unmappable pairs inside every barcode (markDuplicates)
    for bi in range(len(po) - 1):
        lo, hi = int(po[bi]), int(po[bi + 1])
        for _ in range((hi - lo) // 20):
            i, j = rng.integers(lo, hi, size=2)
            rs.seqs[2 * j:2 * j + 2] = rs.seqs[2 * i:2 * i + 2]; rs.lens[2 * j:2 * j + 2] = rs.lens[2 * i:2 * i + 2]
        for j in rng.integers(lo, hi, size=3):
            rs.seqs[2 * j:2 * j + 2] = rng.integers(0, 4, size=rs.seqs[2 * j:2 * j + 2].shape)
    ref = api.load_reference(fa, 0)
    o = oradrv.Oracle(fa)
    S = rs.seqs if (rs.lens == rs.seqs.shape[1]).all() else np.concatenate([rs.seqs[i, :rs.lens[i]] for i in range(len(rs.lens))])   # ragged: flat
    try:
        b = ref.batch(S, rs.lens).run()
        dev = b.fetch()
        ora = o.batch(S, rs.lens, n_threads=8)
        parity.check_final(dev, ora)
        if refdrv.available():
            r = refdrv.Ref(fa); parity.check_final(dev, r.batch(S, rs.lens, n_threads=8))
        po = rs.pair_offsets()
        flags = [rfadrv.worth_running_rfa(rs.barcodes[i], int(po[i + 1] - po[i])) for i in range(len(po) - 1)]
        names, offs, clens, alt, l_pac = ref.contigs()
        orfa = rfadrv.oracle_rfa(ora, rs.lens, po, flags, l_pac, offs)
        parity.check_rfa(b.rfa(po, flags), orfa)
        parity.check_post(b.post(), rfadrv.oracle_post(o.h, ora, S, rs.lens, po, offs, orfa))
        print("seed %d ok: %d regions, %.1fs" % (seed, len(dev["regs"]), time.time() - t), flush=True)
    except AssertionError as e:
        bad += 1
        print("seed %d MISMATCH: %s" % (seed, str(e)[:300]), flush=True)
    ref.close(); o.close()
print("fuzz done, %d mismatching seeds" % bad)
sys.exit(1 if bad else 0)
