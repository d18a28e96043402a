"""Seeded test workloads shared by the golden generator and the parity tests."""
import os
import tempfile

import numpy as np

from arachne_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def nasty_genome(seed, contig_lens=(60000, 30000, 10000), alt_contigs=1):
    """Repeat-rich genome: interspersed families, exact copies, tandem repeats, a homopolymer, N runs."""
    rng = np.random.default_rng(seed)
    g = synth.make_genome(100 + seed, list(contig_lens),
                          repeat_families=[(40, 300, 0.08), (12, 2000, 0.03), (4, 6000, 0.01), (20, 150, 0.0)],
                          alt_contigs=alt_contigs)
    for ci in range(len(contig_lens)):
        s = g.seqs[ci]
        for _ in range(5):
            unit = rng.integers(0, 4, size=int(rng.integers(2, 60)), dtype=np.uint8)
            n = int(rng.integers(5, 40))
            if len(unit) * n + 2 >= len(s):
                continue
            p = int(rng.integers(0, len(s) - len(unit) * n - 1))
            tr = synth._mutate(rng, np.tile(unit, n), 0.02)
            s[p:p + len(tr)] = tr
        p = int(rng.integers(0, len(s) - 300))
        s[p:p + 200] = 0
    return g


def nasty_reads(seed, g, n_barcodes=4, pairs_per_barcode=150):
    """High-error pairs plus corrupted mates (forces rescue), random reads, N runs, chimeras, big indels."""
    rng = np.random.default_rng(1000 + seed)
    rs = synth.make_reads(200 + seed, g, n_barcodes, pairs_per_barcode, sub_rate=0.02, indel_rate=0.004,
                          molecule_len=20000, molecules_per_barcode=3)
    S = rs.seqs
    n = S.shape[0]
    for i in rng.choice(n, size=n // 5, replace=False):
        m = rng.random(150) < 0.12
        S[i, m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
    for i in rng.choice(n, size=max(1, n // 50), replace=False):
        S[i] = rng.integers(0, 4, size=150, dtype=np.uint8)
    for i in rng.choice(n, size=max(1, n // 30), replace=False):
        p = int(rng.integers(0, 140))
        S[i, p:p + int(rng.integers(1, 12))] = 4
    for i in rng.choice(n, size=max(1, n // 30), replace=False):
        j = int(rng.integers(0, n))
        p = int(rng.integers(30, 120))
        S[i, p:] = S[j, p:]
    for i in rng.choice(n, size=max(1, n // 40), replace=False):
        p = int(rng.integers(30, 100))
        L = int(rng.integers(5, 40))
        S[i, p:150 - L] = S[i, p + L:].copy()
    return rs


INDEX_EXTS = ("bwt", "sa", "pac", "ann", "amb", "alt")


def pack_index(prefix):
    """Index files -> dict of uint8 arrays (stored inside the golden .npz)."""
    out = {}
    for ext in INDEX_EXTS:
        fn = prefix + "." + ext
        if os.path.exists(fn):
            out["idx_" + ext] = np.fromfile(fn, dtype=np.uint8)
    return out


def unpack_index(npz, dirname, name="golden.fa"):
    prefix = os.path.join(dirname, name)
    for ext in INDEX_EXTS:
        key = "idx_" + ext
        if key in npz:
            np.asarray(npz[key], dtype=np.uint8).tofile(prefix + "." + ext)
    return prefix


def long_reads(read_len, n_bc=3, ppb=300, mixed=False, contig=1_200_000):
    """Pairs of the given length (up to MAX_READ_LEN = 255), a quarter of the reads corrupted (15 % or 35 % of their bases) so that the rescue SW has work; mixed: every
    third pair keeps only its first 150 bases (a batch that holds both element sizes of ksw_align2: KSW_XBYTE below 250 bases, ksw_i16
    from there on, bwamem_pair.c:150).  Returns genome, ReadSet, sequences (2-D, or flat when mixed) and lengths."""
    from arachne_amd import synth
    g = synth.make_genome(500 + read_len, [contig])
    rs = synth.make_reads(501 + read_len, g, n_bc, ppb, read_len=read_len, sub_rate=0.01)
    rng = np.random.default_rng(read_len)
    for i in rng.choice(rs.seqs.shape[0], size=rs.seqs.shape[0] // 4, replace=False):
        m = rng.random(read_len) < (0.15, 0.35)[i & 1]
        rs.seqs[i, m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
    seqs, lens = rs.seqs, rs.lens
    if mixed:
        lens = lens.copy()
        lens[4::6] = 150
        lens[5::6] = 150
        seqs = np.concatenate([rs.seqs[r, :lens[r]] for r in range(len(lens))])
    return g, rs, seqs, lens
