"""ctypes binding for the Go-half restatement (oracle/arx_oracle_rfa.c: ora_rfa).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORA_SO = os.path.join(HERE, "..", "oracle", "liboracle.so")
CAND_W = 18
CAND_FIELDS = ["reg", "read", "pos", "aend", "reversed", "rid", "score", "mismatches", "indels", "soft_clipped", "soft_clipped_length",
               "lap2", "active", "is_proper", "mapq", "molecule_id", "active_molecule", "in_filtered"]


from arachne_amd.api import worth_running_rfa  # noqa: E402,F401  (host logic of the product, re-exported for the tests)


def oracle_rfa(batch, lens, bc_pair_off, do_rfa, l_pac, ann_off, penalty=-4, centromeres=None):
    """batch: dict with reg_off/regs/alns/cigars in the oracle's int64 row layout."""
    lib = C.CDLL(ORA_SO)
    lib.ora_rfa.restype = C.c_int64
    lib.ora_rfa.argtypes = [C.c_int64] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 5
    n_reads = len(lens)
    reg_off = np.ascontiguousarray(batch["reg_off"], dtype=np.int64)
    regs = np.ascontiguousarray(batch["regs"], dtype=np.int64)
    alns = np.ascontiguousarray(batch["alns"], dtype=np.int64)
    cig = np.ascontiguousarray(batch["cigars"], dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    bco = np.ascontiguousarray(bc_pair_off, dtype=np.int64)
    flags = np.ascontiguousarray(do_rfa, dtype=np.uint8)
    ann = np.ascontiguousarray(ann_off, dtype=np.int64)
    cs = ce = None
    if centromeres is not None:
        cs = np.ascontiguousarray(centromeres[0], dtype=np.int64)
        ce = np.ascontiguousarray(centromeres[1], dtype=np.int64)
    rows = np.zeros((len(regs) + n_reads, CAND_W), dtype=np.int64)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    n = lib.ora_rfa(n_reads, reg_off.ctypes.data, regs.ctypes.data, alns.ctypes.data, cig.ctypes.data, lens.ctypes.data, len(bco) - 1, bco.ctypes.data,
                    flags.ctypes.data, int(penalty), int(l_pac), ann.ctypes.data, cs.ctypes.data if cs is not None else None,
                    ce.ctypes.data if ce is not None else None, rows.ctypes.data, off.ctypes.data)
    return dict(cand_off=off, cands=rows[:n])


POST_FIELDS = ["qb", "qe", "matches", "n_mm", "mm_off", "duplicate"]
SPLIT_FIELDS = ["split", "mapq", "is_proper", "n_split_cand", "order_pinned", "second_best2", "score2"]


def oracle_post(ora_ctx, batch, seqs, lens, bc_pair_off, ann_off, rfa, penalty=-4, centromeres=None):
    """ora_post on the rows of oracle_rfa (`rfa`): CIGAR walk, markDuplicates, CheckSplitReads.  ora_ctx: the ora_ctx_t* of oradrv."""
    lib = C.CDLL(ORA_SO)
    lib.ora_post.restype = C.c_int64
    lib.ora_post.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_int64]
    n_reads = len(lens)
    regs = np.ascontiguousarray(batch["regs"], dtype=np.int64)
    alns = np.ascontiguousarray(batch["alns"], dtype=np.int64)
    cig = np.ascontiguousarray(batch["cigars"], dtype=np.uint32)
    bases = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    bco = np.ascontiguousarray(bc_pair_off, dtype=np.int64)
    ann = np.ascontiguousarray(ann_off, dtype=np.int64)
    cs = ce = None
    if centromeres is not None:
        cs = np.ascontiguousarray(centromeres[0], dtype=np.int64)
        ce = np.ascontiguousarray(centromeres[1], dtype=np.int64)
    rows = np.ascontiguousarray(rfa["cands"], dtype=np.int64)
    off = np.ascontiguousarray(rfa["cand_off"], dtype=np.int64)
    post = np.zeros((len(rows), len(POST_FIELDS)), dtype=np.int64)
    split = np.zeros((n_reads, len(SPLIT_FIELDS)), dtype=np.int64)
    cap = int(lens.sum()) * 4 + 16
    while True:
        mm_ref = np.zeros(cap, dtype=np.int32)
        mm_read = np.zeros(cap, dtype=np.int32)
        n = lib.ora_post(ora_ctx, n_reads, regs.ctypes.data, alns.ctypes.data, cig.ctypes.data, bases.ctypes.data, lens.ctypes.data, len(bco) - 1, bco.ctypes.data,
                         int(penalty), ann.ctypes.data, cs.ctypes.data if cs is not None else None, ce.ctypes.data if ce is not None else None,
                         rows.ctypes.data, off.ctypes.data, post.ctypes.data, split.ctypes.data, mm_ref.ctypes.data, mm_read.ctypes.data, cap)
        if n >= 0:
            break
        cap *= 4
    return dict(post=post, split=split, mm_ref=mm_ref[:n], mm_read=mm_read[:n])
