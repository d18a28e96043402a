"""ctypes binding for oracle/liboracle.so (our CPU restatement, oracle/arx_oracle.c).  It exposes the
same entry points as tests/refdrv.py (ora_* instead of ref_*) so tests can diff the two directly.
Test infrastructure only."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORA_SO = os.path.join(HERE, "..", "oracle", "liboracle.so")

REG_W = 20
ALN_W = 12
REG_FIELDS = ["rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
              "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt", "frac_rep_bits", "pad"]
ALN_FIELDS = ["pos", "rid", "flag", "is_rev", "is_alt", "mapq", "NM", "n_cigar", "cigar_off", "score", "sub", "alt_sc"]


def available() -> bool:
    return os.path.exists(ORA_SO)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, prefix=None):
        self.lib = C.CDLL(ORA_SO)
        L = self.lib
        L.ora_open.restype = C.c_void_p
        L.ora_open.argtypes = [C.c_char_p]
        L.ora_close.argtypes = [C.c_void_p]
        L.ora_l_pac.restype = C.c_int64
        L.ora_l_pac.argtypes = [C.c_void_p]
        L.ora_seq_len.restype = C.c_int64
        L.ora_seq_len.argtypes = [C.c_void_p]
        L.ora_primary.restype = C.c_int64
        L.ora_primary.argtypes = [C.c_void_p]
        L.ora_batch_run.restype = C.c_double
        L.ora_batch_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ora_collect_intv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_chains.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_ksw_extend2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ora_ksw_align2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ora_ksw_global2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_align1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_occ4.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_extend.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ora_sa.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_fetch_seq.restype = C.c_int64
        L.ora_fetch_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.ora_batch_get.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.ora_n_seqs.argtypes = [C.c_void_p]
        self.h = None
        if prefix is not None:
            self.open(prefix)

    def open(self, prefix):
        self.h = self.lib.ora_open(prefix.encode())
        if not self.h:
            raise RuntimeError("ora_open failed for " + prefix)

    def close(self):
        if self.h:
            self.lib.ora_close(self.h)
            self.h = None

    @property
    def l_pac(self):
        return self.lib.ora_l_pac(self.h)

    @property
    def seq_len(self):
        return self.lib.ora_seq_len(self.h)

    @property
    def primary(self):
        return self.lib.ora_primary(self.h)

    # ---- KATs
    def occ4(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros((len(k), 4), dtype=np.uint64)
        self.lib.ora_occ4(self.h, len(k), k.ctypes.data, out.ctypes.data)
        return out

    def extend(self, ik3, is_back):
        ik3 = np.ascontiguousarray(ik3, dtype=np.uint64)
        out = np.zeros((len(ik3), 4, 3), dtype=np.uint64)
        self.lib.ora_extend(self.h, len(ik3), ik3.ctypes.data, int(is_back), out.ctypes.data)
        return out

    def sa(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros(len(k), dtype=np.uint64)
        self.lib.ora_sa(self.h, len(k), k.ctypes.data, out.ctypes.data)
        return out

    def collect_intv(self, seq, cap=4096):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.zeros((cap, 4), dtype=np.uint64)
        n = self.lib.ora_collect_intv(self.h, len(seq), seq.ctypes.data, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def chains(self, seq, do_flt, cap_c=8192, cap_s=65536):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ch = np.zeros((cap_c, 8), dtype=np.int64)
        sd = np.zeros((cap_s, 4), dtype=np.int64)
        ns = C.c_int(0)
        fb = C.c_uint32(0)
        n = self.lib.ora_chains(self.h, len(seq), seq.ctypes.data, int(do_flt), ch.ctypes.data, cap_c, sd.ctypes.data, cap_s, C.byref(ns), C.byref(fb))
        assert n <= cap_c and ns.value <= cap_s
        return ch[:n], sd[:ns.value], fb.value

    def ksw_extend2(self, q, t, w, end_bonus, zdrop, h0):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.zeros(6, dtype=np.int32)
        self.lib.ora_ksw_extend2(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, w, end_bonus, zdrop, h0, out.ctypes.data)
        return out

    def ksw_align2(self, q, t, xtra):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.zeros(7, dtype=np.int32)
        self.lib.ora_ksw_align2(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, xtra, out.ctypes.data)
        return out

    def ksw_align2_i16(self, q, t, xtra):
        """ksw_i16 whatever the flag says (ksw.c:232-334)."""
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.zeros(7, dtype=np.int32)
        self.lib.ora_ksw_align2_i16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        self.lib.ora_ksw_align2_i16(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, xtra, out.ctypes.data)
        return out

    def ksw_global2(self, q, t, w, cap=1024):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        sc = C.c_int(0)
        cg = np.zeros(cap, dtype=np.uint32)
        n = self.lib.ora_ksw_global2(self.h, len(q), q.ctypes.data, len(t), t.ctypes.data, w, C.byref(sc), cg.ctypes.data, cap)
        return sc.value, cg[:n]

    def align1(self, seq, cap=4096):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.zeros((cap, REG_W), dtype=np.int64)
        n = self.lib.ora_align1(self.h, len(seq), seq.ctypes.data, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def fetch_seq(self, beg, mid, end, cap=100000):
        b = C.c_int64(beg)
        e = C.c_int64(end)
        rid = C.c_int(0)
        out = np.zeros(cap, dtype=np.uint8)
        n = self.lib.ora_fetch_seq(self.h, C.byref(b), mid, C.byref(e), C.byref(rid), out.ctypes.data, cap)
        return out[:n], b.value, e.value, rid.value

    # ---- the pair path (gobwa.go:226-337 + 400-415)
    def batch(self, seqs2d_or_flat, lens, score_delta=25, n_threads=1):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        s = np.ascontiguousarray(seqs2d_or_flat, dtype=np.uint8).reshape(-1)
        assert s.size == int(lens.sum())
        n_pairs = len(lens) // 2
        secs = self.lib.ora_batch_run(self.h, n_pairs, s.ctypes.data, lens.ctypes.data, score_delta, n_threads)
        n_reads = C.c_int64()
        n_regs = C.c_int64()
        n_cig = C.c_int64()
        p_off = C.POINTER(C.c_int64)()
        p_regs = C.POINTER(C.c_int64)()
        p_alns = C.POINTER(C.c_int64)()
        p_cig = C.POINTER(C.c_uint32)()
        self.lib.ora_batch_get(self.h, C.byref(n_reads), C.byref(n_regs), C.byref(n_cig), C.byref(p_off), C.byref(p_regs), C.byref(p_alns), C.byref(p_cig))
        nr, nreg, nc = n_reads.value, n_regs.value, n_cig.value
        off = np.ctypeslib.as_array(p_off, shape=(nr + 1,)).copy()
        regs = np.ctypeslib.as_array(p_regs, shape=(max(nreg, 1), REG_W))[:nreg].copy()
        alns = np.ctypeslib.as_array(p_alns, shape=(max(nreg, 1), ALN_W))[:nreg].copy()
        cig = np.ctypeslib.as_array(p_cig, shape=(max(nc, 1),))[:nc].copy()
        return dict(reg_off=off, regs=regs, alns=alns, cigars=cig, secs=secs)

    COUNTER_FIELDS = ["n_reads", "ext_same_block", "ext_two_block", "sa_lookups", "sa_lf_steps", "n_regs",
                      "cells_extend", "cells_u8", "cells_global", "n_extend_calls", "n_u8_calls", "n_global_calls", "ext3_same_block", "ext3_two_block", "extb_same_block", "extb_two_block", "n_smem_calls", "sa_lf_steps8", "sa_lf_steps4", "ext_rows_qlen", "ext_rows_cols"]

    def counters(self, reset=False):
        buf = np.zeros(len(self.COUNTER_FIELDS), dtype=np.int64)
        self.lib.ora_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.lib.ora_counters(self.h, buf.ctypes.data, int(reset))
        return dict(zip(self.COUNTER_FIELDS, buf.tolist()))
