"""Generates tests/golden/bwa_path_v1.npz from the reference's own C core (oracle/_ref/libbwaref.so,
compiled in place from /root/reference by oracle/Makefile).  Runs only where /root/reference exists.

The fixture is DATA: a small synthetic index written by the reference's `bwa index` code path
(bwtindex.c:251-316), seeded reads, and the outputs of the reference routines on them:
  - bwt_occ4 / bwt_extend / bwt_sa at random positions                     (bwt.c:169,262,86)
  - mem_collect_intv interval lists                                         (bwamem.c:114)
  - mem_chain / mem_chain_flt chain dumps                                   (bwamem.c:251,327)
  - ksw_extend2 / ksw_align2 / ksw_global2 on random + adversarial inputs   (ksw.c:380,343,504)
  - mem_align1_core reg lists                                               (bwamem.c:1048)
  - the full per-pair sequence of gobwa.go:226-337 + mem_reg2aln per candidate

usage: python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refdrv  # noqa: E402
import workloads  # noqa: E402

KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


def sw_cases(rng, n):
    """(query, target) pairs: related sequences with substitutions/indels, repeats and ties."""
    out = []
    for i in range(n):
        ql = int(rng.integers(1, 151))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        kind = i % 5
        if kind == 0:      # target = query + noise + flanks
            t = q.copy()
            m = rng.random(ql) < 0.08
            t[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            t = np.concatenate([t, rng.integers(0, 4, size=int(rng.integers(0, 200)), dtype=np.uint8)])
        elif kind == 1:    # indels
            p = int(rng.integers(0, ql))
            L = int(rng.integers(1, 30))
            if rng.random() < 0.5:
                t = np.concatenate([q[:p], rng.integers(0, 4, size=L, dtype=np.uint8), q[p:], rng.integers(0, 4, size=50, dtype=np.uint8)])
            else:
                t = np.concatenate([q[:p], q[min(ql, p + L):], rng.integers(0, 4, size=50, dtype=np.uint8)])
        elif kind == 2:    # low complexity: many exact ties
            unit = rng.integers(0, 4, size=int(rng.integers(1, 4)), dtype=np.uint8)
            q = np.tile(unit, 150)[:ql]
            t = np.tile(unit, 300)[:int(rng.integers(ql, 2 * ql + 20))]
            if rng.random() < 0.5 and len(t) > 10:
                t[int(rng.integers(0, len(t)))] ^= 1
        elif kind == 3:    # unrelated
            t = rng.integers(0, 4, size=int(rng.integers(1, 400)), dtype=np.uint8)
        else:              # with Ns
            t = np.concatenate([q, rng.integers(0, 4, size=30, dtype=np.uint8)])
            t[rng.integers(0, len(t), size=3)] = 4
            q = q.copy()
            q[rng.integers(0, ql)] = 4
        if len(t) == 0:
            t = np.zeros(1, dtype=np.uint8)
        out.append((q, t.astype(np.uint8)))
    return out


def main():
    rng = np.random.default_rng(20250905)
    g = workloads.nasty_genome(7)
    rs = workloads.nasty_reads(7, g, n_barcodes=4, pairs_per_barcode=150)
    tmp = tempfile.mkdtemp(prefix="arx_golden_")
    prefix = os.path.join(tmp, "golden.fa")
    g.write_fasta(prefix)
    g.write_alt(prefix + ".alt")
    r = refdrv.Ref()
    r.index_build(prefix, prefix)
    r.open(prefix)
    d = workloads.pack_index(prefix)
    d["genome_cat"] = np.concatenate(g.seqs)
    d["genome_lens"] = np.array([len(s) for s in g.seqs], dtype=np.int64)
    d["reads"] = rs.seqs
    d["lens"] = rs.lens
    d["barcode_id"] = rs.barcode_id
    # FM-index KATs
    k = rng.integers(0, r.seq_len + 1, size=4000).astype(np.uint64)
    k[:4] = [0, r.primary, r.seq_len, np.uint64(2**64 - 1)]
    d["kat_occ_k"] = k
    d["kat_occ_out"] = r.occ4(k)
    ks = k[4:1004].copy()
    d["kat_sa_k"] = ks
    d["kat_sa_out"] = r.sa(ks)
    # intervals / chains / regs for the first 120 reads
    nint = 120
    iv_off, iv = [0], []
    ch_off, ch, sd_off, sd, fr = [0], [], [0], [], []
    chf_off, chf, sdf_off, sdf = [0], [], [0], []
    rg_off, rg = [0], []
    for i in range(nint):
        a = r.collect_intv(rs.seqs[i])
        iv.append(a)
        iv_off.append(iv_off[-1] + len(a))
        c0, s0, f0 = r.chains(rs.seqs[i], 0)
        ch.append(c0); sd.append(s0); fr.append(f0)
        ch_off.append(ch_off[-1] + len(c0)); sd_off.append(sd_off[-1] + len(s0))
        c1, s1, _ = r.chains(rs.seqs[i], 1)
        chf.append(c1); sdf.append(s1)
        chf_off.append(chf_off[-1] + len(c1)); sdf_off.append(sdf_off[-1] + len(s1))
        x = r.align1(rs.seqs[i])
        rg.append(x)
        rg_off.append(rg_off[-1] + len(x))
    d["kat_intv_off"] = np.array(iv_off); d["kat_intv"] = np.concatenate(iv)
    d["kat_chain_off"] = np.array(ch_off); d["kat_chain"] = np.concatenate(ch)
    d["kat_seed_off"] = np.array(sd_off); d["kat_seed"] = np.concatenate(sd)
    d["kat_fracrep"] = np.array(fr, dtype=np.uint32)
    d["kat_chainf_off"] = np.array(chf_off); d["kat_chainf"] = np.concatenate(chf)
    d["kat_seedf_off"] = np.array(sdf_off); d["kat_seedf"] = np.concatenate(sdf)
    d["kat_reg_off"] = np.array(rg_off); d["kat_reg"] = np.concatenate(rg)
    # SW KATs
    cases = sw_cases(rng, 300)
    qcat = np.concatenate([c[0] for c in cases]); tcat = np.concatenate([c[1] for c in cases])
    d["sw_q"] = qcat; d["sw_t"] = tcat
    d["sw_qlen"] = np.array([len(c[0]) for c in cases], dtype=np.int32)
    d["sw_tlen"] = np.array([len(c[1]) for c in cases], dtype=np.int32)
    ext, aln, glo_sc, glo_cig, glo_off = [], [], [], [], [0]
    ext_par = []
    for i, (q, t) in enumerate(cases):
        w = [100, 200, 5, 37][i % 4]
        h0 = int(rng.integers(1, 120))
        eb = 5
        ext_par.append((w, eb, 100, h0))
        ext.append(r.ksw_extend2(q, t, w, eb, 100, h0))
        aln.append(r.ksw_align2(q, t, KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 19))
        wg = max(abs(len(t) - len(q)) + 3, [3, 10, 50, 100][i % 4])
        sc, cg = r.ksw_global2(q, t, wg)
        glo_sc.append((wg, sc)); glo_cig.append(cg); glo_off.append(glo_off[-1] + len(cg))
    d["sw_ext_par"] = np.array(ext_par, dtype=np.int32); d["sw_ext_out"] = np.array(ext, dtype=np.int32)
    d["sw_aln_out"] = np.array(aln, dtype=np.int32)
    d["sw_glo_par"] = np.array(glo_sc, dtype=np.int32)
    d["sw_glo_cig"] = np.concatenate(glo_cig); d["sw_glo_off"] = np.array(glo_off)
    # the pair path
    out = r.batch(rs.seqs, rs.lens, score_delta=25, n_threads=1)
    d["pair_reg_off"] = out["reg_off"]; d["pair_regs"] = out["regs"]; d["pair_alns"] = out["alns"]; d["pair_cigars"] = out["cigars"]
    path = os.path.join(HERE, "bwa_path_v1.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes; regs", out["regs"].shape[0])


if __name__ == "__main__":
    main()
