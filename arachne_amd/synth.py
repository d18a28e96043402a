"""Synthetic linked-read workloads: genomes with planted repeat families and
paired 2x150 bp haplotagging-style reads grouped by BX barcode.

There is no network on the GPU box, so bench.py and the tests synthesise inputs of the shape
BASELINE.json names.  Everything is seeded (seed = 20250905 + config#) and vectorised with numpy.

Encoding used throughout the package: bases are uint8 0..3 = A,C,G,T and 4 = N (the reference's
nst_nt4_table, /root/reference/src/gobwa/bwa/bntseq.c:47).
"""
from __future__ import annotations

import dataclasses
import numpy as np

BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)


@dataclasses.dataclass
class Genome:
    names: list          # contig names
    seqs: list           # list of uint8 arrays (0..4)
    alt: list            # bool per contig (goes to the .alt file)
    copies: np.ndarray = None   # (n, 4) int64 rows [family, contig, position, length]: where make_genome planted its repeat copies
    fam_weight: np.ndarray = None  # per family: share of the repeat-biased molecules make_reads draws from it

    @property
    def total_len(self) -> int:
        return int(sum(len(s) for s in self.seqs))

    def write_fasta(self, path: str, width: int = 80) -> None:
        with open(path, "wb") as f:
            for name, s in zip(self.names, self.seqs):
                f.write(b">" + name.encode() + b"\n")
                asc = BASES[s]
                n = len(asc)
                full = (n // width) * width
                if full:
                    body = np.empty((n // width, width + 1), dtype=np.uint8)
                    body[:, :width] = asc[:full].reshape(-1, width)
                    body[:, width] = 10
                    f.write(body.tobytes())
                if n > full:
                    f.write(asc[full:].tobytes() + b"\n")

    def write_alt(self, path: str) -> None:
        with open(path, "w") as f:
            for name, a in zip(self.names, self.alt):
                if a:
                    f.write(name + "\n")


def _mutate(rng, s, div):
    s = s.copy()
    m = rng.random(len(s)) < div
    k = int(m.sum())
    if k:
        s[m] = (s[m] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    return s


def segdup_families(n_families: int, weight: float = 1.0):
    """A deterministic list of low-copy segmental-duplication families (2-8 copies of 10-60 kb at 0.5-2.5 % divergence from their
    template, i.e. 1-5 % between copies) for the repeat-enriched workload of BASELINE.json configs[3]: the families real genomes
    have many of, next to the few high-copy ones of SURVEY s8d's recipe.  Rows are make_genome's (copies, length, divergence, weight)."""
    return [(2 + k % 7, 10000 + (k * 7919) % 50000, 0.005 + (k % 5) * 0.005, weight / n_families) for k in range(n_families)]


def make_genome(seed: int, contig_lens, repeat_families=None, n_runs: int = 2, alt_contigs: int = 0, fast: bool = False,
                alt_spec=None, decoy_spec=None) -> Genome:
    """Uniform ACGT contigs + planted repeat families + runs of N.

    repeat_families: list of (copies, length, divergence[, weight]); defaults scale the SURVEY §8d recipe
    (Alu-like 300 bp @12%, L1-like 6 kb @5%, segmental duplications 50 kb @1%) to the genome size.  The optional weight is
    only recorded (make_reads' repeat_bias draws molecules from the families in proportion to it).
    alt_spec = (n, min_len, max_len, divergence): ALT contigs as GRCh38 has them -- diverged copies of slices of the primary
    contigs, flagged in the .alt file (bntseq.c:98-206; rules at bwamem.c:351, 1078-1082); decoy_spec = (n, length): unflagged
    decoy contigs of sequence found nowhere else.  Both come from a random stream of their own, so a genome without them is the
    genome earlier rounds measured.
    """
    rng = np.random.default_rng(seed)
    total = int(sum(contig_lens))
    if fast:  # GRCh38-size genomes: two bits of every random byte (several times faster than bounded integers; a different stream)
        seqs = [np.frombuffer(rng.bytes(int(L)), dtype=np.uint8) & 3 for L in contig_lens]
    else:
        seqs = [rng.integers(0, 4, size=int(L), dtype=np.uint8) for L in contig_lens]
    if repeat_families is None:
        scale = total / 3.1e9
        repeat_families = [
            (max(4, int(1e4 * scale * 20)), 300, 0.12),
            (max(3, int(1e3 * scale * 20)), 6000, 0.05),
            (max(2, int(200 * scale * 10)), 50000, 0.01),
        ]
    planted = []
    for fi, fam in enumerate(repeat_families):
        copies, length, div = fam[:3]
        length = int(min(length, min(contig_lens) // 4))
        if length < 50:
            continue
        tmpl = rng.integers(0, 4, size=length, dtype=np.uint8)
        for _ in range(int(copies)):
            ci = int(rng.integers(0, len(seqs)))
            if len(seqs[ci]) <= length + 2:
                continue
            p = int(rng.integers(0, len(seqs[ci]) - length))
            c = _mutate(rng, tmpl, div)
            if rng.random() < 0.5:
                c = (3 - c)[::-1]
            seqs[ci][p:p + length] = c
            planted.append((fi, ci, p, length))
    for ci in range(len(seqs)):
        for _ in range(n_runs):
            L = len(seqs[ci])
            if L < 5000:
                continue
            ln = int(rng.integers(20, 400))
            p = int(rng.integers(0, L - ln))
            seqs[ci][p:p + ln] = 4
    names = [f"chrS{i + 1}" for i in range(len(seqs))]
    alt = [False] * len(seqs)
    for k in range(alt_contigs):
        # ALT contig = diverged copy of a slice of a primary contig
        src = seqs[k % len(seqs)]
        ln = int(min(20000, len(src) // 3))
        p = int(rng.integers(0, len(src) - ln))
        a = _mutate(rng, src[p:p + ln], 0.02)
        a[a > 3] = 0
        seqs.append(a)
        names.append(f"chrS{(k % len(contig_lens)) + 1}_alt{k + 1}")
        alt.append(True)
    n_prim = len(contig_lens)
    rng2 = np.random.default_rng([seed, 0xA17])
    if alt_spec:
        n_alt, lo, hi, div = alt_spec
        w = np.array([len(seqs[i]) for i in range(n_prim)], dtype=np.float64)
        for k in range(int(n_alt)):
            ci = int(rng2.choice(n_prim, p=w / w.sum()))
            ln = int(min(rng2.integers(lo, hi + 1), len(seqs[ci]) // 3))
            p = int(rng2.integers(0, len(seqs[ci]) - ln))
            a = _mutate(rng2, np.asarray(seqs[ci][p:p + ln]), div)
            a[a > 3] = 0
            seqs.append(a)
            names.append(f"chrS{ci + 1}_KI{270000 + k}v1_alt")
            alt.append(True)
    if decoy_spec:
        n_dec, ln = decoy_spec
        for k in range(int(n_dec)):
            seqs.append(rng2.integers(0, 4, size=int(ln), dtype=np.uint8))
            names.append(f"chrUn_JTFH{1000000 + k}v1_decoy")
            alt.append(False)
    fw = np.array([fam[3] if len(fam) > 3 else 1.0 for fam in repeat_families], dtype=np.float64)
    return Genome(names, seqs, alt, np.array(planted, dtype=np.int64).reshape(-1, 4), fw)


@dataclasses.dataclass
class ReadSet:
    seqs: np.ndarray       # (2*n_pairs, read_len) uint8 0..4; row 2i = R1, 2i+1 = R2
    lens: np.ndarray       # (2*n_pairs,) int32
    barcode_id: np.ndarray  # (n_pairs,) int32 -> index into barcodes
    barcodes: list         # barcode strings (contain '-' so that the reference runs RFA, aligner.go:1022)
    valid: np.ndarray      # (n_pairs,) bool  -> VX:i:1 / VX:i:0
    truth_contig: np.ndarray
    truth_pos: np.ndarray  # leftmost 0-based position of the fragment on the contig

    @property
    def n_pairs(self) -> int:
        return len(self.barcode_id)

    def pair_offsets(self) -> np.ndarray:
        """Offsets of each barcode group in pair units (pairs are stored grouped by barcode)."""
        chg = np.flatnonzero(np.diff(self.barcode_id)) + 1
        return np.concatenate([[0], chg, [self.n_pairs]]).astype(np.int64)


def _revcomp(a):
    r = a[..., ::-1].copy()
    m = r < 4
    r[m] = 3 - r[m]
    return r


def make_reads(seed: int, genome: Genome, n_barcodes: int, pairs_per_barcode: int, read_len: int = 150,
               molecules_per_barcode: int = 10, molecule_len: int = 50000, sub_rate: float = 0.005,
               indel_rate: float = 0.0002, invalid_frac: float = 0.0, repeat_bias=None, fast: bool = False,
               repeat_focus: float = 0.7, barcode_style: str = "haplotag") -> ReadSet:
    """Pairs are FR, insert ~ N(350,50) clipped to >= read_len+10, drawn uniformly inside molecules.
    fast=True (millions of pairs): same distributions, windows gathered as rows of a strided view and substitutions placed by position
    instead of by a per-base mask -- a different random stream, several times quicker.
    repeat_bias (BASELINE.json configs[3]): that fraction of the molecules is drawn from the repeat copies make_genome planted (a
    family in proportion to its weight, a copy of it uniformly; the molecule covers the copy or lies inside it), and repeat_focus of
    their pairs have a mate overlapping the copy -- what makes region and candidate lists long.
    invalid_frac (configs[4]): that fraction of the pairs loses its barcode: VX:i:0, dash-less barcodes shared by 1-4 pairs each,
    filed among the others in barcode order as a barcode-sorted FASTQ has them -- worthRunningRFA is false for every such group
    (aligner.go:1018-1030), they take the fallback of aligner.go:469-477.
    Both draw from random streams of their own: with the defaults the read set is the one earlier rounds measured.
    barcode_style "stlfr": three-part numeric barcodes a_b_c (the '-1' suffix stays, without it the reference never runs RFA)."""
    rng = np.random.default_rng(seed)
    n_pairs = n_barcodes * pairs_per_barcode
    clen = np.array([len(s) for s in genome.seqs], dtype=np.int64)
    prim = np.array([not a for a in genome.alt])
    cw = (clen * prim).astype(np.float64)
    cw /= cw.sum()
    cat = np.concatenate(genome.seqs)
    coff = np.concatenate([[0], np.cumsum(clen)])[:-1]
    # molecules
    n_mol = n_barcodes * molecules_per_barcode
    mol_c = rng.choice(len(clen), size=n_mol, p=cw)
    mlen = np.minimum(molecule_len, clen[mol_c] - 1)
    mol_s = (rng.random(n_mol) * (clen[mol_c] - mlen)).astype(np.int64)
    bc = np.repeat(np.arange(n_barcodes, dtype=np.int32), pairs_per_barcode)
    mol = bc.astype(np.int64) * molecules_per_barcode + rng.integers(0, molecules_per_barcode, size=n_pairs)
    ins = np.clip(rng.normal(350, 50, size=n_pairs).astype(np.int64), read_len + 10, 800)
    ins = np.minimum(ins, mlen[mol] - 1)
    ins = np.maximum(ins, read_len)
    fs = mol_s[mol] + (rng.random(n_pairs) * (mlen[mol] - ins)).astype(np.int64)
    if repeat_bias and genome.copies is not None and len(genome.copies):
        rb = np.random.default_rng([seed, 0xB1A5])
        cp = genome.copies
        fam_w = genome.fam_weight if genome.fam_weight is not None else np.ones(int(cp[:, 0].max()) + 1)
        per_fam = np.bincount(cp[:, 0], minlength=len(fam_w)).astype(np.float64)
        wcopy = fam_w[cp[:, 0]] / per_fam[cp[:, 0]]                      # a family by its weight, then one of its copies uniformly
        biased = np.flatnonzero(rb.random(n_mol) < repeat_bias)
        pick = rb.choice(len(cp), size=len(biased), p=wcopy / wcopy.sum())
        b_c, b_p, b_l = cp[pick, 1], cp[pick, 2], cp[pick, 3]
        b_ml = np.minimum(molecule_len, clen[b_c] - 1)
        # the molecule covers a copy shorter than itself, and lies inside a longer one
        lo = np.where(b_l <= b_ml, b_p + b_l - b_ml, b_p)
        hi = np.where(b_l <= b_ml, b_p, b_p + b_l - b_ml)
        b_s = np.clip(lo + (rb.random(len(biased)) * (hi - lo + 1)).astype(np.int64), 0, clen[b_c] - b_ml)
        mol_c = mol_c.copy(); mol_s = mol_s.copy(); mlen = mlen.copy()
        mol_c[biased], mol_s[biased], mlen[biased] = b_c, b_s, b_ml
        cp_lo = np.full(n_mol, -1, dtype=np.int64); cp_hi = np.zeros(n_mol, dtype=np.int64)
        cp_lo[biased] = np.maximum(b_p, b_s); cp_hi[biased] = np.minimum(b_p + b_l, b_s + b_ml)
        ins = np.maximum(np.minimum(ins, mlen[mol] - 1), read_len)
        fs = mol_s[mol] + (rb.random(n_pairs) * (mlen[mol] - ins)).astype(np.int64)
        foc = np.flatnonzero((cp_lo[mol] >= 0) & (rb.random(n_pairs) < repeat_focus))
        # fragment start so that the fragment overlaps the copy by at least 20 bases, inside the molecule
        f_lo = np.maximum(cp_lo[mol[foc]] - ins[foc] + 20, mol_s[mol[foc]])
        f_hi = np.maximum(np.minimum(cp_hi[mol[foc]] - 20, mol_s[mol[foc]] + mlen[mol[foc]] - ins[foc]), f_lo)
        fs[foc] = f_lo + (rb.random(len(foc)) * (f_hi - f_lo + 1)).astype(np.int64)
    c = mol_c[mol]
    g0 = coff[c] + fs
    flip = rng.random(n_pairs) < 0.5           # which mate is read 1
    seqs = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    if fast:
        win = np.lib.stride_tricks.sliding_window_view(cat, read_len)
        rc = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
        i1 = np.where(flip, 1, 0)              # row (within the pair) of the forward-strand mate
        seqs[2 * np.arange(n_pairs) + i1] = win[g0]
        seqs[2 * np.arange(n_pairs) + 1 - i1] = rc[win[g0 + ins - read_len]][:, ::-1]
        flat = seqs.reshape(-1)
        k = int(rng.binomial(flat.size, sub_rate))
        pos = rng.integers(0, flat.size, size=k)
        pos = pos[flat[pos] < 4]
        flat[pos] = (flat[pos] + rng.integers(1, 4, size=len(pos), dtype=np.uint8)) & 3
    else:
        ar = np.arange(read_len, dtype=np.int64)
        left = cat[g0[:, None] + ar[None, :]]
        right = _revcomp(cat[(g0 + ins - read_len)[:, None] + ar[None, :]])
        r1 = np.where(flip[:, None], right, left)
        r2 = np.where(flip[:, None], left, right)
        seqs[0::2] = r1
        seqs[1::2] = r2
        # substitutions
        m = (rng.random(seqs.shape) < sub_rate) & (seqs < 4)
        k = int(m.sum())
        seqs[m] = (seqs[m] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    # small indels (1-3 bp) on a sparse subset, keeping the read length fixed
    if indel_rate > 0:
        rows = np.flatnonzero(rng.random(2 * n_pairs) < indel_rate * read_len)
        for r in rows:
            p = int(rng.integers(10, read_len - 10))
            ln = int(rng.integers(1, 4))
            s = seqs[r]
            if rng.random() < 0.5:  # deletion from the read
                s[p:read_len - ln] = s[p + ln:]
                s[read_len - ln:] = rng.integers(0, 4, size=ln, dtype=np.uint8)
            else:                   # insertion into the read
                s[p + ln:] = s[p:read_len - ln].copy()
                s[p:p + ln] = rng.integers(0, 4, size=ln, dtype=np.uint8)
    barcodes = []
    alpha = "ABCD"
    for b in range(n_barcodes):
        x = b
        parts = []
        for seg in "ACBD":
            parts.append(f"{seg}{(x % 96) + 1:02d}")
            x //= 96
        barcodes.append("".join(parts) + "-1" if barcode_style != "stlfr" else f"{b % 1536 + 1}_{(b // 1536) % 1536 + 1}_{b // (1536 * 1536) + 1}-1")
    valid = rng.random(n_pairs) >= invalid_frac
    lens = np.full(2 * n_pairs, read_len, dtype=np.int32)
    rs = ReadSet(seqs, lens, bc, barcodes, valid, c.astype(np.int32), fs)
    if invalid_frac > 0:
        rs = _regroup_invalid(np.random.default_rng([seed, 0x1AB]), rs)
    return rs


def _regroup_invalid(rng, rs: ReadSet) -> ReadSet:
    """The pairs flagged invalid leave their barcodes for dash-less ones shared by 1-4 pairs (consecutive invalid pairs in a random
    order, so the pairs of a group come from unrelated molecules), and all groups are put in barcode-string order."""
    inv = np.flatnonzero(~rs.valid)
    inv = inv[rng.permutation(len(inv))]
    sizes = []
    left = len(inv)
    while left > 0:
        k = int(min(left, rng.integers(1, 5)))
        sizes.append(k)
        left -= k
    n_old = len(rs.barcodes)
    grp_of = rs.barcode_id.astype(np.int64).copy()
    grp_of[inv] = n_old + np.repeat(np.arange(len(sizes)), sizes)
    letters = "ABCD"
    new_names = []
    x = rng.integers(0, 96 ** 4, size=len(sizes))
    for g in range(len(sizes)):
        v = int(x[g]); parts = []
        for seg in "ACBD":
            parts.append(f"{seg}{(v % 96) + 1:02d}"); v //= 96
        new_names.append("".join(parts) + f"N{g:06d}")               # no '-': the reference never runs RFA on it, nor emits BX
    names = list(rs.barcodes) + new_names
    order_groups = np.argsort(np.array(names, dtype=object), kind="stable")
    rank = np.empty(len(names), dtype=np.int64); rank[order_groups] = np.arange(len(names))
    perm = np.argsort(rank[grp_of], kind="stable")                   # pairs by group rank, original order inside a group
    used = np.unique(rank[grp_of])                                   # a barcode that lost all its pairs disappears
    remap = np.full(len(names), -1, dtype=np.int64); remap[used] = np.arange(len(used))
    rows = np.stack([2 * perm, 2 * perm + 1], axis=1).reshape(-1)
    return ReadSet(rs.seqs[rows], rs.lens[rows], remap[rank[grp_of[perm]]].astype(np.int32), [names[order_groups[u]] for u in used],
                   rs.valid[perm], rs.truth_contig[perm], rs.truth_pos[perm])


def write_fastq(rs: ReadSet, path1: str, path2: str) -> None:
    """Barcode-sorted FASTQ pair in the header format the reference's reader parses
    (/root/reference/src/fastqreader/reader.go:95-153): '@name/1\\tBX:Z:<bc>\\tVX:i:<0|1>'."""
    with open(path1, "w") as f1, open(path2, "w") as f2:
        for i in range(rs.n_pairs):
            bcs = rs.barcodes[rs.barcode_id[i]]
            vx = 1 if rs.valid[i] else 0
            for f, row, mate in ((f1, 2 * i, 1), (f2, 2 * i + 1, 2)):
                L = int(rs.lens[row])
                s = BASES[rs.seqs[row, :L]].tobytes().decode()
                f.write(f"@r{i}/{mate}\tBX:Z:{bcs}\tVX:i:{vx}\n{s}\n+\n{'I' * L}\n")


def write_fastq_fast(rs: ReadSet, path1: str, path2: str, p0: int = 0, p1: int = None) -> None:
    """write_fastq for millions of pairs: pairs [p0, p1) of the set as two barcode-sorted FASTQ files, records laid out as rows of
    one byte matrix per barcode-string length (same header format: '@r<pair, 9 digits>/<mate>\\tBX:Z:<bc>\\tVX:i:<0|1>'); reads must all
    have the same length."""
    p1 = rs.n_pairs if p1 is None else p1
    n = p1 - p0
    L = int(rs.lens[0])
    assert (rs.lens[2 * p0:2 * p1] == L).all()
    bc_bytes = [b.encode() for b in rs.barcodes]
    bl = np.array([len(b) for b in bc_bytes])
    ids = np.arange(p0, p1)
    digits = ((ids[:, None] // 10 ** np.arange(8, -1, -1)[None, :]) % 10 + 48).astype(np.uint8)
    bid = rs.barcode_id[p0:p1]
    with open(path1, "wb") as f1, open(path2, "wb") as f2:
        # consecutive pairs with the same barcode-string length form one block (all of them, when the barcodes are of one length)
        cut = np.flatnonzero(np.diff(bl[bid])) + 1
        bounds = np.concatenate([[0], cut, [n]])
        for a, b in zip(bounds[:-1], bounds[1:]):
            k = int(bl[bid[a]])
            hl = 1 + 1 + 9 + 2 + 6 + k + 6 + 1 + 1               # @ r digits /m \tBX:Z: bc \tVX:i: v \n
            rec = hl + L + 1 + 2 + L + 1
            bcm = np.frombuffer(b"".join(bc_bytes[i] for i in bid[a:b]), dtype=np.uint8).reshape(b - a, k)
            for f, mate in ((f1, 0), (f2, 1)):
                m = np.empty((b - a, rec), dtype=np.uint8)
                m[:, 0] = 64; m[:, 1] = 114; m[:, 2:11] = digits[a:b]; m[:, 11] = 47; m[:, 12] = 49 + mate
                m[:, 13:19] = np.frombuffer(b"\tBX:Z:", dtype=np.uint8); m[:, 19:19 + k] = bcm
                m[:, 19 + k:25 + k] = np.frombuffer(b"\tVX:i:", dtype=np.uint8); m[:, 25 + k] = 48 + rs.valid[p0 + a:p0 + b]; m[:, 26 + k] = 10
                m[:, hl:hl + L] = BASES[rs.seqs[2 * (p0 + a) + mate:2 * (p0 + b):2, :L]]
                m[:, hl + L] = 10; m[:, hl + L + 1] = 43; m[:, hl + L + 2] = 10
                m[:, hl + L + 3:hl + 2 * L + 3] = 73; m[:, rec - 1] = 10
                f.write(m.tobytes())
