"""Host-side binding of libarachne_amd.so (ctypes over the C ABI of include/arachne_amd.h).

The names mirror the reference's Go bridge (/root/reference/src/gobwa/gobwa.go) so that tests read like its
call sites: `load_reference` ~ GoBwaLoadReference (:128), `Reference.contigs` ~ GetReferenceContigsInfo (:28),
`sequence_convert` ~ SequenceConvert (:159), `Reference.mem_mate_sw` ~ GoBwaMemMateSW (:226) for a whole batch of
pairs followed by GoBwaSmithWaterman (:400) for every candidate.

The library is the product path and it is the only path: if the shared object is missing, or no MI355X is
visible, loading / opening raises -- there is no CPU fallback here.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libarachne_amd.so")

REG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"),
                      ("sub", "<i4"), ("alt_sc", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"), ("seedcov", "<i4"),
                      ("secondary", "<i4"), ("secondary_all", "<i4"), ("seedlen0", "<i4"), ("n_comp", "<i4"), ("is_alt", "<i4"),
                      ("frac_rep", "<f4"), ("pad", "<i4")])
ALN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("flag", "<i4"), ("is_rev", "<i4"), ("is_alt", "<i4"), ("NM", "<i4"),
                      ("n_cigar", "<i4"), ("cigar_off", "<i4"), ("score", "<i4"), ("sub", "<i4"), ("alt_sc", "<i4")])
CHAIN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("n", "<i4"), ("seed_off", "<i4"), ("w", "<i4"), ("kept", "<i4"), ("first", "<i4"),
                        ("is_alt", "<i4"), ("head", "<i4"), ("tail", "<i4"), ("frac_rep", "<f4")])
SEED_DTYPE = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])
CAND_DTYPE = np.dtype([("pos", "<i8"), ("aend", "<i8"), ("sum_move", "<f8"), ("reg", "<i4"), ("read", "<i4"), ("rid", "<i4"), ("reversed", "<i4"),
                       ("score", "<i4"), ("mismatches", "<i4"), ("indels", "<i4"), ("soft_clipped", "<i4"), ("soft_clipped_length", "<i4"),
                       ("lap2", "<i4"), ("active", "<i4"), ("is_proper", "<i4"), ("mapq", "<i4"), ("molecule_id", "<i4"), ("active_molecule", "<i4"),
                       ("in_filtered", "<i4"), ("best_in_mol", "<i4"), ("pad", "<i4")])
CAP_INTV = 256
STAGE_SEED, STAGE_CHAIN, STAGE_EXTEND, STAGE_RESCUE, STAGE_ALN = 1, 2, 3, 4, 5

POST_DTYPE = np.dtype([("qb", "<i4"), ("qe", "<i4"), ("matches", "<i4"), ("n_mm", "<i4"), ("mm_off", "<i4"), ("duplicate", "<i4")])
SPLIT_DTYPE = np.dtype([("split", "<i4"), ("mapq", "<i4"), ("is_proper", "<i4"), ("n_split_cand", "<i4"), ("order_pinned", "<i4"),
                        ("second_best2", "<i4"), ("score2", "<i4"), ("pad", "<i4")])
_NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _NT4[ord(_c)] = _i
    _NT4[ord(_c.lower())] = _i


def sequence_convert(seq) -> np.ndarray:
    """ASCII bases -> codes 0..4 (nst_nt4_table; '-' is not special-cased on this path)."""
    if isinstance(seq, str):
        seq = seq.encode()
    return _NT4[np.frombuffer(seq, dtype=np.uint8)]


class ArachneError(RuntimeError):
    pass


def _load(path):
    if not os.path.exists(path):
        raise ArachneError(f"{path} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.arx_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.arx_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, i32]
    lib.arx_close.argtypes = [vp]
    lib.arx_last_error.restype = C.c_char_p
    lib.arx_last_error.argtypes = [vp]
    lib.arx_backend.restype = C.c_char_p
    lib.arx_index_info.argtypes = [vp, vp]
    lib.arx_host_register.argtypes = [vp, C.c_int64]
    lib.arx_batch_detach.argtypes = [vp, vp, vp]
    lib.arx_batch_fetch_detached.argtypes = [vp] * 8
    lib.arx_host_unregister.argtypes = [vp]
    lib.arx_contigs.argtypes = [vp] + [vp] * 6
    lib.arx_batch_create.argtypes = [vp, i32, vp, vp, C.POINTER(vp)]
    lib.arx_batch_reset.argtypes = [vp, vp, i32, vp, vp]
    lib.arx_batch_reset_device.argtypes = [vp, vp, i32, i64, vp, vp]
    lib.arx_batch_device_view.argtypes = [vp, vp, vp]
    lib.arx_batch_run.argtypes = [vp, vp, i32]
    lib.arx_batch_counts.argtypes = [vp, vp, vp]
    lib.arx_batch_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.arx_batch_free.argtypes = [vp, vp]
    lib.arx_batch_debug_intv.argtypes = [vp, vp, vp, vp]
    lib.arx_batch_debug_chains.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.arx_batch_debug_core.argtypes = [vp, vp, vp, vp]
    lib.arx_batch_rfa.argtypes = [vp, vp, i32, vp, vp, C.c_double, vp, vp, vp]
    lib.arx_batch_rfa_fetch.argtypes = [vp, vp, vp, vp]
    lib.arx_batch_post.argtypes = [vp, vp, vp]
    lib.arx_batch_post_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.arx_feeder_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp), C.c_char_p, i32]
    lib.arx_feeder_next.argtypes = [vp, i64, vp]
    lib.arx_feeder_close.argtypes = [vp]
    lib.arx_bam_open.argtypes = [C.c_char_p, i32, vp, vp, C.c_char_p, i32, i32, C.POINTER(vp), C.c_char_p, i32]
    lib.arx_bam_write.argtypes = [vp, vp]
    lib.arx_bam_close.argtypes = [vp, vp]
    lib.arx_bam_error.restype = C.c_char_p
    lib.arx_bam_error.argtypes = [vp]
    lib.arx_recbuf_create.argtypes = [C.POINTER(vp)]
    lib.arx_recbuf_build.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.arx_recbuf_error.restype = C.c_char_p
    lib.arx_recbuf_error.argtypes = [vp]
    lib.arx_recbuf_free.argtypes = [vp]
    lib.arx_multi_open.argtypes = [C.c_char_p, i32, vp, C.POINTER(vp), C.c_char_p, i32]
    lib.arx_multi_run.argtypes = [vp, i32, vp, vp, i32, vp, vp, C.c_double, vp, vp, vp]
    lib.arx_multi_error.restype = C.c_char_p
    lib.arx_multi_error.argtypes = [vp]
    lib.arx_multi_close.argtypes = [vp]
    lib.arx_kernel_times.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    lib.arx_kernel_times_reset.argtypes = [vp, i32]
    lib.arx_selftest_wave_sort.argtypes = [i32, i32, C.c_int64, vp]
    return lib


def selftest_wave_sort(n_cases: int, seed: int = 1, device: int = 0, lib_path: str = LIB_PATH) -> int:
    """klib's introsort as the wavefront kernels reproduce it against the one-thread original on random index arrays -> arrays that differ."""
    lib = _load(lib_path)
    bad = C.c_int64(-1)
    rc = lib.arx_selftest_wave_sort(device, n_cases, seed, C.byref(bad))
    if rc != 0:
        raise ArachneError("arx_selftest_wave_sort: code %d" % rc)
    return int(bad.value)


def index_build(fasta: str, prefix: str, lib_path: str = LIB_PATH) -> None:
    """`bwa index` equivalent (host side): writes <prefix>.{bwt,sa,pac,ann,amb}, byte-identical to the reference's."""
    lib = _load(lib_path)
    msg = C.create_string_buffer(512)
    if lib.arx_index_build(fasta.encode(), prefix.encode(), msg, 512) != 0:
        raise ArachneError("arx_index_build: " + msg.value.decode())


class _DeviceView(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_regs", C.c_int64), ("n_cigar", C.c_int64), ("n_cands", C.c_int64), ("reg_off", C.c_void_p), ("regs", C.c_void_p),
                ("alns", C.c_void_p), ("cigars", C.c_void_p), ("cand_off", C.c_void_p), ("cands", C.c_void_p)]


class Batch:
    """One batch of read pairs resident on the device (arx_batch)."""

    def __init__(self, ref: "Reference", seqs, lens):
        self.ref = ref
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        bases = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
        if bases.size != int(lens.sum()):
            raise ArachneError("bases do not match lens")
        self.n_reads = len(lens)
        self._keep = (bases, lens)
        self._n_cands = 0
        h = C.c_void_p()
        ref._check(ref.lib.arx_batch_create(ref.h, self.n_reads, bases.ctypes.data, lens.ctypes.data, C.byref(h)))
        self.h = h
        ref._batches.add(self)

    def reset(self, seqs, lens):
        """New reads into the same handle (arx_batch_reset): stream, work memory and input buffers are reused."""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        bases = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
        self.n_reads = len(lens)
        self._keep = (bases, lens)
        self._n_cands = 0
        self.ref._check(self.ref.lib.arx_batch_reset(self.ref.h, self.h, self.n_reads, bases.ctypes.data, lens.ctypes.data))
        return self

    def reset_device(self, n_reads: int, n_bases: int, d_bases: int, d_lens: int):
        """arx_batch_reset_device: new reads from DEVICE pointers (e.g. a tensor received over RCCL); the caller's stream must be done with them."""
        self.n_reads = int(n_reads)
        self._keep = None
        self._n_cands = 0
        self.ref._check(self.ref.lib.arx_batch_reset_device(self.ref.h, self.h, int(n_reads), int(n_bases), C.c_void_p(d_bases), C.c_void_p(d_lens)))
        return self

    def device_view(self):
        """arx_batch_device_view -> dict name -> (device pointer, element count, numpy dtype) of the dense result arrays where they lie"""
        v = _DeviceView()
        self.ref._check(self.ref.lib.arx_batch_device_view(self.ref.h, self.h, C.byref(v)))
        out = dict(reg_off=(v.reg_off, v.n_reads + 1, np.dtype(np.int32)), regs=(v.regs, v.n_regs, REG_DTYPE), alns=(v.alns, v.n_regs, ALN_DTYPE),
                   cigars=(v.cigars, v.n_cigar, np.dtype(np.uint32)))
        if v.cands:
            out["cand_off"] = (v.cand_off, v.n_reads + 1, np.dtype(np.int32))
            out["cands"] = (v.cands, v.n_cands, CAND_DTYPE)
        return out

    def pin(self, arr):
        """arx_host_register on a numpy array the caller keeps reusing (inputs of reset(), buffers of fetch_into()); released when the array is
        collected.  Returns the array; silently leaves it pageable when the registration is refused."""
        import weakref
        if arr.nbytes and self.ref.lib.arx_host_register(arr.ctypes.data, arr.nbytes) == 0:
            weakref.finalize(arr, self.ref.lib.arx_host_unregister, arr.ctypes.data)
        return arr

    def fetch_into(self, buf):
        """arx_batch_fetch + arx_batch_rfa_fetch into arrays the caller keeps (buf: dict with reg_off, regs, alns, cigars, cand_off, cands,
        each at least as long as this batch needs; grown here when not): what a steady-state caller does, no allocation per batch."""
        c = self.counts()
        need = dict(reg_off=(self.n_reads + 1, np.int32), regs=(c["n_regs"], REG_DTYPE), alns=(c["n_regs"], ALN_DTYPE), cigars=(max(c["n_cigar"], 1), np.uint32),
                    cand_off=(self.n_reads + 1, np.int32), cands=(self._n_cands, CAND_DTYPE))
        for k, (n, dt) in need.items():
            if k not in buf or len(buf[k]) < n:
                buf[k] = None                                            # (the old array's finalizer unregisters it)
                buf[k] = self.pin(np.zeros(int(n * 1.2) + 16, dtype=dt))   # page-locked: the results arrive by DMA, no staging copy
        self.ref._check(self.ref.lib.arx_batch_fetch(self.ref.h, self.h, buf["reg_off"].ctypes.data, buf["regs"].ctypes.data, buf["alns"].ctypes.data, buf["cigars"].ctypes.data))
        if self._n_cands:
            self.ref._check(self.ref.lib.arx_batch_rfa_fetch(self.ref.h, self.h, buf["cand_off"].ctypes.data, buf["cands"].ctypes.data))
        return c

    def detach(self):
        """arx_batch_detach: the results copied aside on the device; the handle is free for reset() / run() while another thread calls
        fetch_detached_into().  -> sizes dict"""
        a = np.zeros(4, dtype=np.int64)
        self.ref._check(self.ref.lib.arx_batch_detach(self.ref.h, self.h, a.ctypes.data))
        return dict(n_reads=int(a[0]), n_regs=int(a[1]), n_cigar=int(a[2]), n_cands=int(a[3]))

    def fetch_detached_into(self, buf, sizes):
        """arx_batch_fetch_detached into arrays the caller keeps (grown and page-locked here when needed), from any thread."""
        need = dict(reg_off=(sizes["n_reads"] + 1, np.int32), regs=(sizes["n_regs"], REG_DTYPE), alns=(sizes["n_regs"], ALN_DTYPE), cigars=(max(sizes["n_cigar"], 1), np.uint32),
                    cand_off=(sizes["n_reads"] + 1, np.int32), cands=(max(sizes["n_cands"], 1), CAND_DTYPE))
        for k, (n, dt) in need.items():
            if k not in buf or len(buf[k]) < n:
                buf[k] = None
                buf[k] = self.pin(np.zeros(int(n * 1.2) + 16, dtype=dt))
        self.ref._check(self.ref.lib.arx_batch_fetch_detached(self.ref.h, self.h, buf["reg_off"].ctypes.data, buf["regs"].ctypes.data, buf["alns"].ctypes.data,
                                                              buf["cigars"].ctypes.data, buf["cand_off"].ctypes.data, buf["cands"].ctypes.data))
        return sizes

    def post_into(self, buf):
        """arx_batch_post + arx_batch_post_fetch of the per-candidate records into buf["post"] (grown when needed)."""
        n = C.c_int64()
        self.ref._check(self.ref.lib.arx_batch_post(self.ref.h, self.h, C.byref(n)))
        if "post" not in buf or len(buf["post"]) < self._n_cands:
            buf["post"] = None
            buf["post"] = self.pin(np.zeros(int(self._n_cands * 1.2) + 16, dtype=POST_DTYPE))
        self.ref._check(self.ref.lib.arx_batch_post_fetch(self.ref.h, self.h, buf["post"].ctypes.data, None, None, None))
        return buf["post"]

    def run(self, last_stage=STAGE_ALN):
        self.ref._check(self.ref.lib.arx_batch_run(self.ref.h, self.h, last_stage))
        return self

    def counts(self):
        c = np.zeros(8, dtype=np.int64)
        self.ref._check(self.ref.lib.arx_batch_counts(self.ref.h, self.h, c.ctypes.data))
        return dict(zip(["n_reads", "n_regs", "n_cigar", "n_occ", "ext_rounds", "n_ext", "rescue_rounds", "n_sw"], c.tolist()))

    def fetch(self):
        c = self.counts()
        reg_off = np.zeros(self.n_reads + 1, dtype=np.int32)
        regs = np.zeros(c["n_regs"], dtype=REG_DTYPE)
        alns = np.zeros(c["n_regs"], dtype=ALN_DTYPE)
        cig = np.zeros(max(c["n_cigar"], 1), dtype=np.uint32)
        self.ref._check(self.ref.lib.arx_batch_fetch(self.ref.h, self.h, reg_off.ctypes.data, regs.ctypes.data, alns.ctypes.data, cig.ctypes.data))
        return dict(reg_off=reg_off, regs=regs, alns=alns, cigars=cig[:c["n_cigar"]], counts=c)

    def debug_intv(self):
        n = np.zeros(self.n_reads, dtype=np.int32)
        iv = np.zeros((self.n_reads, CAP_INTV, 4), dtype=np.uint64)
        self.ref._check(self.ref.lib.arx_batch_debug_intv(self.ref.h, self.h, n.ctypes.data, iv.ctypes.data))
        return n, iv

    def debug_chains(self):
        T = self.counts()["n_occ"]
        off = np.zeros(self.n_reads + 1, dtype=np.int32)
        n = np.zeros(self.n_reads, dtype=np.int32)
        ch = np.zeros(max(T, 1), dtype=CHAIN_DTYPE)
        sd = np.zeros(max(T, 1), dtype=SEED_DTYPE)
        self.ref._check(self.ref.lib.arx_batch_debug_chains(self.ref.h, self.h, off.ctypes.data, n.ctypes.data, ch.ctypes.data, sd.ctypes.data))
        return off, n, ch, sd

    def debug_core(self):
        T = self.counts()["n_occ"]
        n = np.zeros(self.n_reads, dtype=np.int32)
        rg = np.zeros(max(T, 1), dtype=REG_DTYPE)
        self.ref._check(self.ref.lib.arx_batch_debug_core(self.ref.h, self.h, n.ctypes.data, rg.ctypes.data))
        return n, rg

    def rfa(self, bc_pair_off, do_rfa, penalty=-4, centromeres=None, fetch=True):
        """The Go half for this batch (needs run(STAGE_ALN)): per-barcode joint placement and MAPQ.
        -> dict(cand_off, cands) with one record per candidate (see arx_cand); fetch=False leaves the records in the library."""
        bco = np.ascontiguousarray(bc_pair_off, dtype=np.int64)
        flags = np.ascontiguousarray(do_rfa, dtype=np.uint8)
        cs = ce = None
        if centromeres is not None:
            cs = np.ascontiguousarray(centromeres[0], dtype=np.int64)
            ce = np.ascontiguousarray(centromeres[1], dtype=np.int64)
        n = C.c_int64()
        self.ref._check(self.ref.lib.arx_batch_rfa(self.ref.h, self.h, len(bco) - 1, bco.ctypes.data, flags.ctypes.data, float(penalty),
                                                   cs.ctypes.data if cs is not None else None, ce.ctypes.data if ce is not None else None, C.byref(n)))
        self._n_cands = int(n.value)
        if not fetch:
            return int(n.value)
        off = np.zeros(self.n_reads + 1, dtype=np.int32)
        cands = np.zeros(n.value, dtype=CAND_DTYPE)
        self.ref._check(self.ref.lib.arx_batch_rfa_fetch(self.ref.h, self.h, off.ctypes.data, cands.ctypes.data))
        return dict(cand_off=off, cands=cands)

    def post(self, fetch=True):
        """The passes between placement and the BAM records (needs rfa()): CIGAR walk with mismatch locations, markDuplicates,
        CheckSplitReads.  -> dict(post, split, mm_ref, mm_read), see arx_cand_post / arx_split."""
        n = C.c_int64()
        self.ref._check(self.ref.lib.arx_batch_post(self.ref.h, self.h, C.byref(n)))
        if not fetch:
            return int(n.value)
        post = np.zeros(self._n_cands, dtype=POST_DTYPE)
        split = np.zeros(self.n_reads, dtype=SPLIT_DTYPE)
        mm_ref = np.zeros(max(n.value, 1), dtype=np.int32)
        mm_read = np.zeros(max(n.value, 1), dtype=np.int32)
        self.ref._check(self.ref.lib.arx_batch_post_fetch(self.ref.h, self.h, post.ctypes.data, split.ctypes.data, mm_ref.ctypes.data, mm_read.ctypes.data))
        return dict(post=post, split=split, mm_ref=mm_ref[:n.value], mm_read=mm_read[:n.value])

    def free(self):
        if self.h:
            self.ref.lib.arx_batch_free(self.ref.h, self.h)
            self.h = None
            self.ref._batches.discard(self)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _SuperBatch(C.Structure):
    _fields_ = [("n_sets", C.c_int32), ("pad", C.c_int32), ("n_pairs", C.c_int64), ("bad_lines", C.c_int64), ("set_pair_off", C.c_void_p),
                ("unique", C.c_void_p), ("do_rfa", C.c_void_p), ("bases", C.c_void_p), ("quals", C.c_void_p), ("lens", C.c_void_p),
                ("valid", C.c_void_p), ("name_off", C.c_void_p), ("names", C.c_void_p), ("rg_off", C.c_void_p), ("rgs", C.c_void_p),
                ("barcode_off", C.c_void_p), ("barcodes", C.c_void_p)]


class Feeder:
    """The reference's paired FASTQ reader (fastqreader.OpenFastQ / ReadBarcodeSet) delivering super-batches of whole barcode sets
    (arx_feeder_*).  Host code of the product library; needs no GPU."""

    def __init__(self, r1: str, r2: str, lib_path: str = LIB_PATH):
        self.lib = _load(lib_path)
        self.h = C.c_void_p()
        msg = C.create_string_buffer(512)
        if self.lib.arx_feeder_open(r1.encode(), r2.encode(), C.byref(self.h), msg, 512) != 0:
            raise ArachneError("arx_feeder_open: " + msg.value.decode())

    def next(self, target_pairs: int):
        """-> dict (numpy copies) or None at the end of the input"""
        sb = _SuperBatch()
        n = self.lib.arx_feeder_next(self.h, int(target_pairs), C.byref(sb))
        if n < 0:
            raise ArachneError("arx_feeder_next: read error")
        if n == 0:
            return None

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dtype=dt)
            return np.frombuffer(C.string_at(ptr, count * np.dtype(dt).itemsize), dtype=dt).copy()

        P = sb.n_pairs
        lens = arr(sb.lens, 2 * P, np.int32)
        nb = int(lens.sum())
        name_off, rg_off, bc_off = arr(sb.name_off, P + 1, np.int64), arr(sb.rg_off, P + 1, np.int64), arr(sb.barcode_off, n + 1, np.int64)
        names, rgs, bcs = C.string_at(sb.names, int(name_off[-1])), C.string_at(sb.rgs, int(rg_off[-1])), C.string_at(sb.barcodes, int(bc_off[-1]))
        return dict(n_sets=n, n_pairs=P, bad_lines=sb.bad_lines, set_pair_off=arr(sb.set_pair_off, n + 1, np.int64), unique=arr(sb.unique, n, np.uint8),
                    do_rfa=arr(sb.do_rfa, n, np.uint8), bases=arr(sb.bases, nb, np.uint8), quals=C.string_at(sb.quals, nb), lens=lens,
                    valid=arr(sb.valid, P, np.uint8),
                    names=[names[name_off[i]:name_off[i + 1]].decode() for i in range(P)],
                    rgs=[rgs[rg_off[i]:rg_off[i + 1]].decode() for i in range(P)],
                    barcodes=[bcs[bc_off[i]:bc_off[i + 1]].decode() for i in range(n)])

    def next_raw(self, target_pairs: int):
        """-> (_SuperBatch, views) without copying, or None at the end of the input.  The struct goes to RecBuf.build as it is; views are
        numpy arrays over the feeder's own memory (bases, lens, set_pair_off, do_rfa: what arx_batch_create / arx_batch_reset and
        arx_batch_rfa take).  Everything is valid until the next call on this feeder."""
        sb = _SuperBatch()
        n = self.lib.arx_feeder_next(self.h, int(target_pairs), C.byref(sb))
        if n < 0:
            raise ArachneError("arx_feeder_next: read error")
        if n == 0:
            return None

        def view(ptr, count, ct, dt):
            return np.frombuffer((ct * count).from_address(ptr), dtype=dt) if count else np.zeros(0, dtype=dt)
        P = sb.n_pairs
        lens = view(sb.lens, 2 * P, C.c_int32, np.int32)
        nb = int(lens.sum(dtype=np.int64))
        return sb, dict(n_sets=n, n_pairs=P, lens=lens, bases=view(sb.bases, nb, C.c_uint8, np.uint8), set_pair_off=view(sb.set_pair_off, n + 1, C.c_int64, np.int64),
                        do_rfa=view(sb.do_rfa, n, C.c_uint8, np.uint8))

    def close(self):
        if self.h:
            self.lib.arx_feeder_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BamBatch(C.Structure):
    _fields_ = [("n_records", C.c_int64), ("name_off", C.c_void_p), ("names", C.c_void_p), ("flag", C.c_void_p), ("rid", C.c_void_p), ("pos", C.c_void_p),
                ("mapq", C.c_void_p), ("mate_rid", C.c_void_p), ("mate_pos", C.c_void_p), ("tlen", C.c_void_p), ("cigar_off", C.c_void_p), ("cigars", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p), ("qual_offset", C.c_int32), ("aux_off", C.c_void_p), ("aux", C.c_void_p)]


class BamWriter:
    """The BAM sink behind the path (arx_bam_*): records in batches of flat arrays, encoded and BGZF-compressed on `threads` host threads.
    Host code of the product library; needs no GPU."""

    def __init__(self, path: str, contig_names, contig_lens, extra_header: str = "", threads: int = 8, level: int = -1, lib_path: str = LIB_PATH):
        self.lib = _load(lib_path)
        self.h = C.c_void_p()
        n = len(contig_names)
        names = (C.c_char_p * n)(*[x.encode() for x in contig_names])
        lens = np.ascontiguousarray(contig_lens, dtype=np.int32)
        msg = C.create_string_buffer(512)
        if self.lib.arx_bam_open(path.encode(), n, names, lens.ctypes.data, extra_header.encode() if extra_header else None, threads, level, C.byref(self.h), msg, 512) != 0:
            raise ArachneError("arx_bam_open: " + msg.value.decode())

    def write(self, names, flag, rid, pos, mapq, mate_rid, mate_pos, tlen, cigars, seqs, quals, aux, qual_offset=33):
        """names / seqs / quals / aux: lists of bytes; cigars: list of uint32 arrays (BAM words); the rest arrays of length n."""
        n = len(names)
        def cat(parts, dt=np.uint8):
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(p) for p in parts])
            flat = np.concatenate([np.frombuffer(p, dtype=np.uint8) if isinstance(p, (bytes, bytearray)) else np.asarray(p, dtype=dt) for p in parts]) if n and off[-1] else np.zeros(1, dtype=dt)
            return off, np.ascontiguousarray(flat, dtype=dt)
        name_off, name_b = cat(names)
        cig_off, cig_w = cat(cigars, np.uint32)
        seq_off, seq_b = cat(seqs)
        _q, qual_b = cat(quals)
        aux_off, aux_b = cat(aux)
        keep = [np.ascontiguousarray(x, dtype=dt) for x, dt in ((flag, np.int32), (rid, np.int32), (pos, np.int32), (mapq, np.uint8), (mate_rid, np.int32), (mate_pos, np.int32), (tlen, np.int32))]
        b = _BamBatch(n, name_off.ctypes.data, name_b.ctypes.data, keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data, keep[4].ctypes.data,
                      keep[5].ctypes.data, keep[6].ctypes.data, cig_off.ctypes.data, cig_w.ctypes.data, seq_off.ctypes.data, seq_b.ctypes.data, qual_b.ctypes.data, qual_offset,
                      aux_off.ctypes.data, aux_b.ctypes.data)
        if self.lib.arx_bam_write(self.h, C.byref(b)) != 0:
            raise ArachneError("arx_bam_write: " + self.lib.arx_bam_error(self.h).decode())

    def write_view(self, view):
        """a _BamBatch as RecBuf.build returns it"""
        if self.lib.arx_bam_write(self.h, C.byref(view)) != 0:
            raise ArachneError("arx_bam_write: " + self.lib.arx_bam_error(self.h).decode())

    def close(self):
        st = np.zeros(4, dtype=np.int64)
        if self.h:
            rc = self.lib.arx_bam_close(self.h, st.ctypes.data)
            self.h = C.c_void_p()
            if rc != 0:
                raise ArachneError("arx_bam_close failed")
        return dict(records=int(st[0]), blocks=int(st[1]), bytes_in=int(st[2]), bytes_out=int(st[3]))


class RecBuf:
    """From the path's results to BAM records (arx_recbuf_*): the primary record of every read of a super-batch, built on host threads;
    the view it returns goes to BamWriter.write_view.  Host code of the product library."""

    def __init__(self, lib_path: str = LIB_PATH):
        self.lib = _load(lib_path)
        self.h = C.c_void_p()
        if self.lib.arx_recbuf_create(C.byref(self.h)) != 0:
            raise ArachneError("arx_recbuf_create failed")

    def build(self, sb, cand_off, cands, alns, cigars, post=None, threads: int = 8):
        """sb: the _SuperBatch of Feeder.next_raw; the arrays as Batch.fetch_into / Batch.post leave them -> _BamBatch view"""
        view = _BamBatch()
        rc = self.lib.arx_recbuf_build(self.h, C.byref(sb), cand_off.ctypes.data, cands.ctypes.data, alns.ctypes.data, cigars.ctypes.data,
                                       post.ctypes.data if post is not None else None, int(threads), C.byref(view))
        if rc != 0:
            raise ArachneError("arx_recbuf_build: " + self.lib.arx_recbuf_error(self.h).decode())
        return view

    def free(self):
        if self.h:
            self.lib.arx_recbuf_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _MultiResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_regs", C.c_int64), ("n_cigar", C.c_int64), ("n_cands", C.c_int64), ("reg_off", C.c_void_p), ("regs", C.c_void_p),
                ("alns", C.c_void_p), ("cigars", C.c_void_p), ("cand_off", C.c_void_p), ("cands", C.c_void_p), ("device_of_barcode", C.c_void_p)]


class MultiReference:
    """Several GPUs behind one handle (arx_multi_*): whole barcodes assigned by pair count, one host thread per device, results in the
    order of the read set -- what a Go caller binds to drive a node."""

    def __init__(self, prefix: str, devices, lib_path: str = LIB_PATH):
        self.lib = _load(lib_path)
        self.h = C.c_void_p()
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        msg = C.create_string_buffer(512)
        if self.lib.arx_multi_open(prefix.encode(), len(dv), dv.ctypes.data, C.byref(self.h), msg, 512) != 0:
            self.h = None
            raise ArachneError("arx_multi_open: " + msg.value.decode())

    def run(self, seqs, lens, bc_pair_off, do_rfa, penalty=-4):
        """-> dict(reg_off, regs, alns, cigars, cand_off, cands, device_of_barcode) as numpy copies"""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        bases = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
        bco = np.ascontiguousarray(bc_pair_off, dtype=np.int64)
        flags = np.ascontiguousarray(do_rfa, dtype=np.uint8)
        r = _MultiResult()
        if self.lib.arx_multi_run(self.h, len(lens), bases.ctypes.data, lens.ctypes.data, len(bco) - 1, bco.ctypes.data, flags.ctypes.data, float(penalty), None, None, C.byref(r)) != 0:
            raise ArachneError("arx_multi_run: " + self.lib.arx_multi_error(self.h).decode())

        def arr(ptr, n, dt):
            return np.frombuffer(C.string_at(ptr, int(n) * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dtype=dt)
        return dict(reg_off=arr(r.reg_off, r.n_reads + 1, np.int32), regs=arr(r.regs, r.n_regs, REG_DTYPE), alns=arr(r.alns, r.n_regs, ALN_DTYPE),
                    cigars=arr(r.cigars, r.n_cigar, np.uint32), cand_off=arr(r.cand_off, r.n_reads + 1, np.int32), cands=arr(r.cands, r.n_cands, CAND_DTYPE),
                    device_of_barcode=arr(r.device_of_barcode, len(bco) - 1, np.int32))

    def close(self):
        if self.h:
            self.lib.arx_multi_close(self.h)
            self.h = None


def worth_running_rfa(barcode: str, n_pairs: int, unique: bool = True) -> bool:
    """worthRunningRFA (aligner.go:1018-1030): the barcode came through unique, has a '-' in it, and holds at least 5 pairs."""
    return bool(unique and n_pairs >= 5 and len(barcode.split("-")) >= 2)


class Reference:
    """A loaded index resident in HBM (arx_ctx); mirrors gobwa.GoBwaReference + GoBwaSettings."""

    def __init__(self, prefix: str, device: int = 0, lib_path: str = LIB_PATH):
        self.lib = _load(lib_path)
        self.h = C.c_void_p()
        rc = self.lib.arx_open(prefix.encode(), device, C.byref(self.h))
        if rc != 0:
            msg = self.lib.arx_last_error(None).decode()
            self.h = None
            raise ArachneError(f"arx_open({prefix}) failed: {msg}")
        self.backend = self.lib.arx_backend().decode()
        self._batches = weakref.WeakSet()   # batches alive on this context (arx_close frees what is left)

    def _check(self, rc):
        if rc != 0:
            raise ArachneError(f"libarachne_amd error {rc}: {self.lib.arx_last_error(self.h).decode()}")

    def contigs(self):
        """-> (names, offsets, lengths, is_alt, l_pac)"""
        n = C.c_int32()
        names = C.POINTER(C.c_char_p)()
        offs = C.POINTER(C.c_int64)()
        lens = C.POINTER(C.c_int32)()
        alt = C.POINTER(C.c_int32)()
        lp = C.c_int64()
        self._check(self.lib.arx_contigs(self.h, C.byref(n), C.byref(names), C.byref(offs), C.byref(lens), C.byref(alt), C.byref(lp)))
        k = n.value
        return ([names[i].decode() for i in range(k)], [offs[i] for i in range(k)], [lens[i] for i in range(k)], [alt[i] for i in range(k)], lp.value)

    def index_info(self) -> dict:
        """What arx_open built beside the files' content (arx_index_info)."""
        a = np.zeros(8, dtype=np.int64)
        self._check(self.lib.arx_index_info(self.h, a.ctypes.data))
        return dict(symbols=int(a[0]), kmer_k=int(a[1]), kmer_fwd_depth=int(a[2]), sa_rows_per_entry=int(a[3]), text_mode=bool(a[4]), device_bytes=int(a[5]))

    def batch(self, seqs, lens) -> Batch:
        return Batch(self, seqs, lens)

    def mem_mate_sw(self, seqs, lens):
        """Whole hot path for a batch of pairs (rows 2i / 2i+1 are mates): candidate regions of both reads after
        mate rescue and the alignment record (pos, strand, NM, CIGAR) of every candidate."""
        b = self.batch(seqs, lens)
        try:
            return b.run().fetch()
        finally:
            b.free()

    def kernel_times(self, cap=64):
        names = C.create_string_buffer(cap * 32)
        ms = np.zeros(cap, dtype=np.float64)
        calls = np.zeros(cap, dtype=np.int64)
        items = np.zeros(cap, dtype=np.int64)
        n = self.lib.arx_kernel_times(self.h, cap, names, 32, ms.ctypes.data, calls.ctypes.data, items.ctypes.data)
        out = {}
        for i in range(n):
            nm = names.raw[i * 32:(i + 1) * 32].split(b"\0")[0].decode()
            out[nm] = dict(ms=float(ms[i]), calls=int(calls[i]), items=int(items[i]))
        return out

    def kernel_times_reset(self, enable=True):
        self.lib.arx_kernel_times_reset(self.h, int(enable))

    def close(self):
        if self.h:
            for b in list(self._batches):      # arx_close frees them: make sure no Python object frees them again
                b.h = None
            self._batches.clear()
            self.lib.arx_close(self.h)
            self.h = None


def load_reference(prefix: str, device: int = 0) -> Reference:
    # ARX_LIB (experiments only): another build of the same HIP library, e.g. a kernel variant under comparison
    return Reference(prefix, device, lib_path=os.environ.get("ARX_LIB", LIB_PATH))
