"""The rescue pre-filter (arachne_amd/csrc/dev_sw.h: sw_prefilter_serial; the GPU kernel k_sw_filter_g16 computes the same sum):
a task it drops must score below min_seed_len = 19 in ksw_align2, which is all mem_matesw asks (bwamem_pair.c:153).
Checked here against the oracle's ksw_align2 on constructions that sit on the bound's worst cases -- alignments of score 19..30
made of short runs separated by mismatches, with and without gaps -- and on random windows; every rescue alignment of every
workload of the suite goes through the same check inside the host test double (ARX_SW_FILTER_CHECK aborts on a violation)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

import oradrv
import workloads

SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libarx_hostsim.so")
XTRA = 0x40000 | 0x80000 | 0x10000 | 19     # KSW_XSUBO | KSW_XSTART | KSW_XBYTE | min_seed_len, as mem_matesw passes it


def _mosaic(rng, t, qlen):
    """a query holding a copy of a piece of t cut into runs of 5..9 matches by mismatches, sometimes with a gap: score near 19"""
    q = rng.integers(0, 4, size=qlen).astype(np.uint8)
    n_runs = int(rng.integers(1, 8))
    runs = rng.integers(5, 10, size=n_runs)
    if n_runs == 1:
        runs[0] = int(rng.integers(17, 24))
    L = int(runs.sum() + n_runs - 1)
    tp = int(rng.integers(0, len(t) - L - 40))
    qp = int(rng.integers(0, qlen - L - 1))
    piece = t[tp:tp + L].copy()
    at = 0
    for r in runs[:-1]:
        at += int(r)
        piece[at] = (piece[at] + 1 + rng.integers(0, 3)) & 3      # a mismatch between two runs
        at += 1
    if rng.random() < 0.3 and L > 20:                             # a deletion or an insertion in the middle
        cut = int(rng.integers(8, L - 8))
        g = int(rng.integers(1, 4))
        piece = np.concatenate([piece[:cut], piece[cut + g:]]) if rng.random() < 0.5 else np.concatenate([piece[:cut], rng.integers(0, 4, size=g).astype(np.uint8), piece[cut:]])
    q[qp:qp + len(piece)] = piece[:qlen - qp]
    return q


def test_filtered_tasks_score_below_min_seed_len(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    sim = C.CDLL(SIM)
    sim.arx_test_sw_prefilter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bwa_path_v1.npz"))
    o = oradrv.Oracle(workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_swf_")))      # any index: ksw_align2 only needs the scoring matrix
    rng = np.random.default_rng(11)
    n_low = n_filtered = n_high = 0
    for it in range(6000):
        kind = it % 4
        tlen = int(rng.integers(60 if kind == 0 else 200, 785))
        qlen = int(rng.integers(30 if kind == 0 else 100, 250))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        if kind == 0:
            q = rng.integers(0, 4, size=qlen).astype(np.uint8)                         # nothing planted
        elif kind == 3:
            q = np.tile(rng.integers(0, 4, size=int(rng.integers(1, 4))).astype(np.uint8), qlen)[:qlen]   # low complexity against ...
            t[rng.integers(0, tlen - 30):][:30] = q[:30]                               # ... a window holding a stretch of it
        else:
            q = _mosaic(rng, t, qlen)
        if it % 50 == 7:
            q[int(rng.integers(0, qlen))] = 4                                          # an N: never filtered
        passed = sim.arx_test_sw_prefilter(q.ctypes.data, qlen, t.ctypes.data, tlen)
        score = int(o.ksw_align2(q, t, XTRA)[0])
        if 4 in q:
            assert passed
        if not passed:
            assert score < 19, (it, score)
            n_filtered += 1
        n_low += score < 19
        n_high += score >= 19
    assert n_high > 1500 and n_low > 1500          # both sides of the threshold are exercised ...
    assert n_filtered > 0.7 * n_low                 # ... and the filter catches most of what it may catch


def _sw_pairs(rng, n):
    """(query, target) pairs that make F work: indels of every length on both sides, low-complexity runs with exact ties, two copies of the query,
    unrelated sequence, Ns -- queries of 20..249 bases, windows of up to 700."""
    out = []
    for i in range(n):
        ql = int(rng.integers(20, 250))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        kind = i % 6
        if kind == 0:
            t = q.copy()
            for _ in range(int(rng.integers(1, 4))):                                   # several indels
                p = int(rng.integers(0, len(t)))
                L = int(rng.integers(1, 25))
                t = np.concatenate([t[:p], rng.integers(0, 4, size=L, dtype=np.uint8), t[p:]]) if rng.random() < 0.5 else np.concatenate([t[:p], t[min(len(t), p + L):]])
            t = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 150)), dtype=np.uint8), t, rng.integers(0, 4, size=int(rng.integers(0, 150)), dtype=np.uint8)])
        elif kind == 1:
            unit = rng.integers(0, 4, size=int(rng.integers(1, 5)), dtype=np.uint8)
            q = np.tile(unit, 260)[:ql]
            t = np.tile(unit, 800)[:int(rng.integers(ql, min(700, 3 * ql)))].copy()
            for _ in range(int(rng.integers(0, 4))):
                t[int(rng.integers(0, len(t)))] ^= 1
        elif kind == 2:
            c2 = q.copy()
            m = rng.random(ql) < 0.04
            c2[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            t = np.concatenate([q[: ql // 2], rng.integers(0, 4, size=int(rng.integers(1, 40)), dtype=np.uint8), q[ql // 2:], rng.integers(0, 4, size=int(rng.integers(5, 90)), dtype=np.uint8), c2])
        elif kind == 3:
            t = rng.integers(0, 4, size=int(rng.integers(30, 700)), dtype=np.uint8)
        elif kind == 4:
            t = np.concatenate([q, rng.integers(0, 4, size=30, dtype=np.uint8)])
            t[rng.integers(0, len(t), size=3)] = 4
            q = q.copy(); q[int(rng.integers(0, ql))] = 4
        else:
            m = rng.random(ql) < 0.12
            t = q.copy(); t[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            t = np.concatenate([rng.integers(0, 4, size=40, dtype=np.uint8), t])
        out.append((q, np.ascontiguousarray(t[:700], dtype=np.uint8)))
    return out


def test_plain_recurrence_for_f_gives_what_the_lazy_loop_gives(built):
    """The rescue kernel computes a row's F by a prefix scan (hip_sw_coop.h) where ksw_u8 runs its lazy-F loop after a striped main pass that
    restarts F at every stripe and feeds E the H from before the loop.  The H values -- hence score, ends, second-best score and its end, starts
    -- are the same: checked here with the plain recurrence (dev_sw.h u8_pass, exact_f) against the restatement of the reference's procedure on
    pairs built to exercise F (and against the golden vectors the compiled reference wrote)."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    lib = C.CDLL(SIM)
    z = np.load(os.path.join(workloads.GOLDEN_DIR, "bwa_path_v1.npz"))
    o = oradrv.Oracle(workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_swx_")))

    lib.arx_test_sw_exact_f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def dev(q, t, xtra, exact):
        out = np.zeros(7, dtype=np.int32)
        lib.arx_test_sw_exact_f(q.ctypes.data, len(q), t.ctypes.data, len(t), xtra, exact, out.ctypes.data)
        return out

    qo = np.concatenate([[0], np.cumsum(z["sw_qlen"])]); to = np.concatenate([[0], np.cumsum(z["sw_tlen"])])
    for i in range(len(z["sw_qlen"])):                                                   # the reference's own outputs
        q = np.ascontiguousarray(z["sw_q"][qo[i]:qo[i + 1]]); t = np.ascontiguousarray(z["sw_t"][to[i]:to[i + 1]])
        assert (dev(q, t, XTRA, 1) == z["sw_aln_out"][i]).all(), i
    rng = np.random.default_rng(4242)
    n_gapped = 0
    for i, (q, t) in enumerate(_sw_pairs(rng, 420)):
        for xtra in (XTRA, 0x40000 | 0x10000 | 30):
            exp = o.ksw_align2(q, t, xtra)
            assert (dev(q, t, xtra, 0) == exp).all(), ("lazy", i)
            assert (dev(q, t, xtra, 1) == exp).all(), ("plain", i, dev(q, t, xtra, 1), exp)
        n_gapped += int(exp[0] >= 19)
    assert n_gapped > 200
