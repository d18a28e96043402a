"""Generates tests/golden/ksw_i16_v1.npz from the reference's own C core (oracle/_ref/libbwaref.so, compiled in place from
/root/reference by oracle/Makefile).  Runs only where /root/reference exists.

The fixture is DATA for the 16-bit element size of ksw_align2 (ksw.c:232-334), which mem_matesw takes for mates of 250 bases and
more (bwamem_pair.c:150: `xtra = KSW_XSUBO | KSW_XSTART | (l_ms * a < 250 ? KSW_XBYTE : 0) | min_seed_len * a`):
  - ksw_align2 WITHOUT KSW_XBYTE on random + adversarial (query, target) pairs, queries of 1..255 bases;
  - the per-pair path (gobwa.go:226-337 + mem_reg2aln per candidate) on 250- and 255-base pairs with corrupted mates, on a small
    synthetic index the reference's own `bwa index` path wrote (the index is rebuilt by the tests from the stored contigs).

usage: python tests/golden/make_i16_golden.py
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

KSW_XSUBO, KSW_XSTART = 0x40000, 0x80000


def sw_cases(rng, n):
    out = []
    for i in range(n):
        ql = int(rng.integers(1, 256)) if i % 3 else int(rng.integers(250, 256))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        kind = i % 6
        if kind == 0:      # target = query + noise + flank
            t = q.copy()
            m = rng.random(ql) < 0.06
            t[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            t = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 120)), dtype=np.uint8), t, rng.integers(0, 4, size=int(rng.integers(0, 200)), dtype=np.uint8)])
        elif kind == 1:    # one indel
            p = int(rng.integers(0, ql))
            L = int(rng.integers(1, 30))
            if rng.random() < 0.5:
                t = np.concatenate([q[:p], rng.integers(0, 4, size=L, dtype=np.uint8), q[p:], rng.integers(0, 4, size=50, dtype=np.uint8)])
            else:
                t = np.concatenate([q[:p], q[min(ql, p + L):], rng.integers(0, 4, size=50, dtype=np.uint8)])
        elif kind == 2:    # low complexity: exact ties, scores above 255 would need the wide elements
            unit = rng.integers(0, 4, size=int(rng.integers(1, 4)), dtype=np.uint8)
            q = np.tile(unit, 255)[:ql]
            t = np.tile(unit, 700)[:int(rng.integers(ql, 2 * ql + 20))]
            if rng.random() < 0.5 and len(t) > 10:
                t[int(rng.integers(0, len(t)))] ^= 1
        elif kind == 3:    # unrelated
            t = rng.integers(0, 4, size=int(rng.integers(1, 600)), dtype=np.uint8)
        elif kind == 4:    # two copies of the query in the target: second-best score and its end (KSW_XSUBO)
            c2 = q.copy()
            m = rng.random(ql) < 0.03
            c2[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            t = np.concatenate([q, rng.integers(0, 4, size=int(rng.integers(5, 90)), dtype=np.uint8), c2])
        else:              # with Ns
            t = np.concatenate([q, rng.integers(0, 4, size=30, dtype=np.uint8)])
            t[rng.integers(0, len(t), size=3)] = 4
            q = q.copy()
            q[rng.integers(0, ql)] = 4
        out.append((q, np.ascontiguousarray(t, dtype=np.uint8)))
    return out


def main():
    rng = np.random.default_rng(20261005)
    d = {}
    tmp = tempfile.mkdtemp(prefix="arx_i16_golden_")
    r = refdrv.Ref()
    # the scoring matrix is all ksw_align2 needs: any index will do, take the pair-path one
    g, rs, seqs, lens = workloads.long_reads(250, n_bc=2, ppb=100, mixed=True, contig=150_000)
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    r.index_build(prefix, prefix)
    r.open(prefix)
    cases = sw_cases(rng, 360)
    d["sw_q"] = np.concatenate([c[0] for c in cases])
    d["sw_t"] = np.concatenate([c[1] for c in cases])
    d["sw_qlen"] = np.array([len(c[0]) for c in cases], dtype=np.int32)
    d["sw_tlen"] = np.array([len(c[1]) for c in cases], dtype=np.int32)
    d["sw_xtra"] = np.array([KSW_XSUBO | KSW_XSTART | 19, KSW_XSUBO | 19, KSW_XSTART | 30, 0][:4], dtype=np.int32)
    d["sw_out"] = np.array([[r.ksw_align2(q, t, int(x)) for x in d["sw_xtra"]] for q, t in cases], dtype=np.int32)
    # pair path, mixed 250 / 150 and plain 255
    for tag, (gg, rr, ss, ll) in (("mix250", (g, rs, seqs, lens)), ("all255", workloads.long_reads(255, n_bc=2, ppb=80, contig=150_000))):
        if tag != "mix250":
            r.close()
            prefix = os.path.join(tmp, tag + ".fa")
            gg.write_fasta(prefix)
            r.index_build(prefix, prefix)
            r.open(prefix)
        out = r.batch(ss, ll, n_threads=2)
        d[tag + "_contig"] = gg.seqs[0]
        d[tag + "_reads"] = np.ascontiguousarray(ss).reshape(-1)
        d[tag + "_lens"] = ll
        for k in ("reg_off", "regs", "alns", "cigars"):
            d[tag + "_" + k] = out[k]
    r.close()
    path = os.path.join(HERE, "ksw_i16_v1.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes;", len(cases), "SW cases;", {t: int(d[t + "_regs"].shape[0]) for t in ("mix250", "all255")})


if __name__ == "__main__":
    main()
