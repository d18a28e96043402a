"""Shared parity checks: device-path results (arachne_amd.api, any backend exporting the C ABI) vs the oracle."""
import numpy as np

from arachne_amd import api

REG_COLS = ["rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
            "secondary_all", "seedlen0", "n_comp", "is_alt"]
ALN_MAP = [("pos", 0), ("rid", 1), ("flag", 2), ("is_rev", 3), ("is_alt", 4), ("NM", 6), ("n_cigar", 7), ("score", 9), ("sub", 10), ("alt_sc", 11)]


def regs_to_rows(regs):
    """arx_reg records -> the oracle's 20-column int64 rows"""
    out = np.zeros((len(regs), 20), dtype=np.int64)
    for i, c in enumerate(REG_COLS):
        out[:, i] = regs[c]
    out[:, 18] = regs["frac_rep"].view(np.uint32)
    return out


def check_final(dev, ora):
    """dev: Batch.fetch() dict; ora: oradrv/refdrv batch() dict.  Raises AssertionError with the first differing read."""
    assert (dev["reg_off"].astype(np.int64) == ora["reg_off"]).all(), "region counts per read differ"
    rows = regs_to_rows(dev["regs"])
    if not (rows == ora["regs"]).all():
        bad = np.argwhere(rows != ora["regs"])[0]
        raise AssertionError(f"region row {bad[0]} col {bad[1]}: dev {rows[bad[0]]} ora {ora['regs'][bad[0]]}")
    for name, col in ALN_MAP:
        a, b = dev["alns"][name].astype(np.int64), ora["alns"][:, col]
        if not (a == b).all():
            i = int(np.argwhere(a != b)[0][0])
            raise AssertionError(f"aln field {name} differs at region {i}: dev {a[i]} ora {b[i]}")
    # CIGARs region by region (offsets are batch-relative in both)
    assert (dev["alns"]["cigar_off"].astype(np.int64) == ora["alns"][:, 8]).all(), "cigar offsets differ"
    assert dev["cigars"].shape == ora["cigars"].shape and (dev["cigars"] == ora["cigars"]).all(), "cigar words differ"


def check_intervals(batch, ora, seqs, lens, reads=None):
    n, iv = batch.debug_intv()
    off = np.concatenate([[0], np.cumsum(lens)])
    flat = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
    for r in (range(len(lens)) if reads is None else reads):
        exp = ora.collect_intv(flat[off[r]:off[r + 1]]) if lens[r] >= 19 else np.zeros((0, 4), dtype=np.uint64)
        got = iv[r, :n[r]]
        assert got.shape == exp.shape and (got == exp).all(), f"intervals differ for read {r}"


def check_chains(batch, ora, seqs, lens, reads=None):
    occ_off, n_chain, ch, sd = batch.debug_chains()
    off = np.concatenate([[0], np.cumsum(lens)])
    flat = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
    for r in (range(len(lens)) if reads is None else reads):
        ec, es, fr = ora.chains(flat[off[r]:off[r + 1]], 1)
        g0 = occ_off[r]
        c = ch[g0:g0 + n_chain[r]]
        assert len(c) == len(ec), f"chain count differs for read {r}: {len(c)} vs {len(ec)}"
        for i in range(len(c)):
            assert (c["pos"][i], c["rid"][i], c["n"][i], c["w"][i], c["kept"][i], c["is_alt"][i]) == tuple(ec[i][[0, 1, 2, 4, 5, 7]]), (r, i)
            s = sd[c["seed_off"][i]:c["seed_off"][i] + c["n"][i]]
            e = es[ec[i][3]:ec[i][3] + ec[i][2]]
            assert (s["rbeg"] == e[:, 0]).all() and (s["qbeg"] == e[:, 1]).all() and (s["len"] == e[:, 2]).all(), (r, i)
        if len(c):
            assert c["frac_rep"][:1].view(np.uint32)[0] == fr


def check_core(batch, ora, seqs, lens, reads=None):
    n_core, rg = batch.debug_core()
    occ_off = batch.debug_chains()[0]
    off = np.concatenate([[0], np.cumsum(lens)])
    flat = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
    for r in (range(len(lens)) if reads is None else reads):
        exp = ora.align1(flat[off[r]:off[r + 1]])
        got = regs_to_rows(rg[occ_off[r]:occ_off[r] + n_core[r]])
        assert got.shape == exp.shape and (got == exp).all(), f"core regions differ for read {r}:\n{got}\n{exp}"


RFA_FIELDS = [("reg", 0), ("read", 1), ("pos", 2), ("aend", 3), ("reversed", 4), ("rid", 5), ("score", 6), ("mismatches", 7), ("indels", 8),
              ("soft_clipped", 9), ("soft_clipped_length", 10), ("lap2", 11), ("active", 12), ("is_proper", 13), ("mapq", 14),
              ("molecule_id", 15), ("active_molecule", 16), ("in_filtered", 17)]


def check_rfa(dev, ora):
    """dev: Batch.rfa() dict; ora: rfadrv.oracle_rfa() dict.  Every candidate field, bit for bit (MAPQ only on active candidates)."""
    assert (dev["cand_off"].astype(np.int64) == ora["cand_off"]).all(), "candidate counts per read differ"
    d, o = dev["cands"], ora["cands"]
    assert len(d) == len(o)
    for name, col in RFA_FIELDS:
        a, b = d[name].astype(np.int64), o[:, col]
        if name == "mapq":
            m = o[:, 12] == 1
            a, b = a[m], b[m]
        if not (a == b).all():
            i = int(np.argwhere(a != b)[0][0])
            raise AssertionError(f"candidate field {name} differs at row {i}: dev {a[i]} ora {b[i]}")


def check_post(dev, ora):
    """dev: Batch.post() dict; ora: rfadrv.oracle_post() dict.  Per candidate: qb/qe, matches, mismatch lists, duplicate; per read: the split record."""
    import rfadrv
    for col, name in enumerate(rfadrv.POST_FIELDS):
        a, b = dev["post"][name].astype(np.int64), ora["post"][:, col]
        if not (a == b).all():
            i = int(np.argwhere(a != b)[0][0])
            raise AssertionError(f"post field {name} differs at candidate {i}: dev {a[i]} ora {b[i]}")
    assert len(dev["mm_ref"]) == len(ora["mm_ref"])
    assert (dev["mm_ref"] == ora["mm_ref"]).all(), "mismatch reference locations differ"
    assert (dev["mm_read"] == ora["mm_read"]).all(), "mismatch read locations differ"
    for col, name in enumerate(rfadrv.SPLIT_FIELDS):
        a, b = dev["split"][name].astype(np.int64), ora["split"][:, col]
        if not (a == b).all():
            i = int(np.argwhere(a != b)[0][0])
            raise AssertionError(f"split field {name} differs at read {i}: dev {a[i]} ora {b[i]}")
