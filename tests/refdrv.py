"""ctypes binding for oracle/_ref/libbwaref.so (the reference's own C core compiled in place by
oracle/Makefile).  Test infrastructure only."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "..", "oracle", "_ref", "libbwaref.so")

REG_W = 20
ALN_W = 12
REG_FIELDS = ["rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
              "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt", "frac_rep_bits", "pad"]
ALN_FIELDS = ["pos", "rid", "flag", "is_rev", "is_alt", "mapq", "NM", "n_cigar", "cigar_off", "score", "sub", "alt_sc"]


def available() -> bool:
    return os.path.exists(REF_SO)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Ref:
    def __init__(self, prefix=None):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.ref_open.restype = C.c_void_p
        L.ref_open.argtypes = [C.c_char_p]
        L.ref_close.argtypes = [C.c_void_p]
        L.ref_index_build.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_l_pac.restype = C.c_int64
        L.ref_l_pac.argtypes = [C.c_void_p]
        L.ref_seq_len.restype = C.c_int64
        L.ref_seq_len.argtypes = [C.c_void_p]
        L.ref_primary.restype = C.c_int64
        L.ref_primary.argtypes = [C.c_void_p]
        L.ref_batch_run.restype = C.c_double
        L.ref_batch_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ref_collect_intv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_chains.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_ksw_extend2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_ksw_align2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_ksw_global2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_align1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_occ4.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_extend.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_sa.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_fetch_seq.restype = C.c_int64
        L.ref_fetch_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_batch_get.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.ref_n_seqs.argtypes = [C.c_void_p]
        L.ref_phase_split.restype = C.c_double
        L.ref_phase_split.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.h = None
        if prefix is not None:
            self.open(prefix)

    def index_build(self, fasta, prefix):
        return self.lib.ref_index_build(fasta.encode(), prefix.encode())

    def open(self, prefix):
        self.h = self.lib.ref_open(prefix.encode())
        if not self.h:
            raise RuntimeError("ref_open failed for " + prefix)

    def close(self):
        if self.h:
            self.lib.ref_close(self.h)
            self.h = None

    @property
    def l_pac(self):
        return self.lib.ref_l_pac(self.h)

    @property
    def seq_len(self):
        return self.lib.ref_seq_len(self.h)

    @property
    def primary(self):
        return self.lib.ref_primary(self.h)

    # ---- KATs
    def occ4(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros((len(k), 4), dtype=np.uint64)
        self.lib.ref_occ4(self.h, len(k), k.ctypes.data, out.ctypes.data)
        return out

    def extend(self, ik3, is_back):
        ik3 = np.ascontiguousarray(ik3, dtype=np.uint64)
        out = np.zeros((len(ik3), 4, 3), dtype=np.uint64)
        self.lib.ref_extend(self.h, len(ik3), ik3.ctypes.data, int(is_back), out.ctypes.data)
        return out

    def sa(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros(len(k), dtype=np.uint64)
        self.lib.ref_sa(self.h, len(k), k.ctypes.data, out.ctypes.data)
        return out

    def collect_intv(self, seq, cap=4096):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.zeros((cap, 4), dtype=np.uint64)
        n = self.lib.ref_collect_intv(self.h, len(seq), seq.ctypes.data, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def chains(self, seq, do_flt, cap_c=8192, cap_s=65536):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ch = np.zeros((cap_c, 8), dtype=np.int64)
        sd = np.zeros((cap_s, 4), dtype=np.int64)
        ns = C.c_int(0)
        fb = C.c_uint32(0)
        n = self.lib.ref_chains(self.h, len(seq), seq.ctypes.data, int(do_flt), ch.ctypes.data, cap_c, sd.ctypes.data, cap_s, C.byref(ns), C.byref(fb))
        assert n <= cap_c and ns.value <= cap_s
        return ch[:n], sd[:ns.value], fb.value

    def ksw_extend2(self, q, t, w, end_bonus, zdrop, h0):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.zeros(6, dtype=np.int32)
        self.lib.ref_ksw_extend2(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, w, end_bonus, zdrop, h0, out.ctypes.data)
        return out

    def ksw_align2(self, q, t, xtra):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.zeros(7, dtype=np.int32)
        self.lib.ref_ksw_align2(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, xtra, out.ctypes.data)
        return out

    def ksw_global2(self, q, t, w, cap=1024):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        sc = C.c_int(0)
        cg = np.zeros(cap, dtype=np.uint32)
        n = self.lib.ref_ksw_global2(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, w, C.byref(sc), cg.ctypes.data, cap)
        return sc.value, cg[:n]

    def align1(self, seq, cap=4096):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.zeros((cap, REG_W), dtype=np.int64)
        n = self.lib.ref_align1(self.h, len(seq), seq.ctypes.data, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def fetch_seq(self, beg, mid, end, cap=100000):
        b = C.c_int64(beg)
        e = C.c_int64(end)
        rid = C.c_int(0)
        out = np.zeros(cap, dtype=np.uint8)
        n = self.lib.ref_fetch_seq(self.h, C.byref(b), mid, C.byref(e), C.byref(rid), out.ctypes.data, cap)
        return out[:n], b.value, e.value, rid.value

    # ---- the pair path (gobwa.go:226-337 + 400-415)
    def batch(self, seqs2d_or_flat, lens, score_delta=25, n_threads=1):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        s = np.ascontiguousarray(seqs2d_or_flat, dtype=np.uint8).reshape(-1)
        assert s.size == int(lens.sum())
        n_pairs = len(lens) // 2
        secs = self.lib.ref_batch_run(self.h, n_pairs, s.ctypes.data, lens.ctypes.data, score_delta, n_threads)
        n_reads = C.c_int64()
        n_regs = C.c_int64()
        n_cig = C.c_int64()
        p_off = C.POINTER(C.c_int64)()
        p_regs = C.POINTER(C.c_int64)()
        p_alns = C.POINTER(C.c_int64)()
        p_cig = C.POINTER(C.c_uint32)()
        self.lib.ref_batch_get(self.h, C.byref(n_reads), C.byref(n_regs), C.byref(n_cig), C.byref(p_off), C.byref(p_regs), C.byref(p_alns), C.byref(p_cig))
        nr, nreg, nc = n_reads.value, n_regs.value, n_cig.value
        off = np.ctypeslib.as_array(p_off, shape=(nr + 1,)).copy()
        regs = np.ctypeslib.as_array(p_regs, shape=(max(nreg, 1), REG_W))[:nreg].copy()
        alns = np.ctypeslib.as_array(p_alns, shape=(max(nreg, 1), ALN_W))[:nreg].copy()
        cig = np.ctypeslib.as_array(p_cig, shape=(max(nc, 1),))[:nc].copy()
        return dict(reg_off=off, regs=regs, alns=alns, cigars=cig, secs=secs)

    def phase_split(self, seqs2d_or_flat, lens, score_delta=25, n_threads=1):
        """Thread-seconds of the reference C core per phase on these pairs (timing only; results are discarded):
        -> dict(seed, extend, rescue, cigar, n_regs, wall)."""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        s = np.ascontiguousarray(seqs2d_or_flat, dtype=np.uint8).reshape(-1)
        out = np.zeros(5, dtype=np.float64)
        wall = self.lib.ref_phase_split(self.h, len(lens) // 2, s.ctypes.data, lens.ctypes.data, score_delta, n_threads, out.ctypes.data)
        return dict(seed=float(out[0]), extend=float(out[1]), rescue=float(out[2]), cigar=float(out[3]), n_regs=int(out[4]), wall=float(wall))
