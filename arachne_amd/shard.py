"""Barcode-sharded multi-GPU dataflow: one read set at an ingest rank, whole barcodes assigned to GPUs, packed
batches scattered, result slabs gathered -- the host-side counterpart of the reference's producer loop, which hands one WorkUnit
(one barcode set of fastqreader.ReadBarcodeSet, at most 30,000 pairs) to one worker at a time (src/aligner/aligner.go:335-358).

Barcode groups are independent (DoRFAForOneBarcode touches only its WorkUnit, aligner.go:440-501) and the index is replicated, so the
only exchange is this scatter and gather; there is no all-reduce anywhere.  torch.distributed point-to-point carries it
(backend "nccl" = RCCL over xGMI between GPUs, "gloo" in the CPU tests): per destination one header and one byte payload.

What the gather returns at rank 0 is, byte for byte, what ONE device batch over the whole read set returns at N = 1
(tests/test_shard_scatter.py): reads, regions and CIGAR words are renumbered into the read set's own order.
"""
from __future__ import annotations

import numpy as np

from . import api


def lpt_assign(pair_counts, n_ranks):
    """Greedy longest-processing-time assignment of whole barcodes to ranks by pair count: barcodes by decreasing size (ties: lower
    index first), each to the rank with the least pairs so far (ties: lower rank).  -> list of barcode-index arrays, each in
    increasing barcode order (a rank processes its barcodes in the order of the read set)."""
    pair_counts = np.asarray(pair_counts, dtype=np.int64)
    order = np.lexsort((np.arange(len(pair_counts)), -pair_counts))
    load = np.zeros(n_ranks, dtype=np.int64)
    mine = [[] for _ in range(n_ranks)]
    for b in order:
        r = int(np.argmin(load))          # first minimum: the lower rank
        mine[r].append(int(b))
        load[r] += pair_counts[b]
    return [np.array(sorted(m), dtype=np.int64) for m in mine]


def pack(seqs, lens, pair_off, do_rfa, barcodes):
    """The reads of `barcodes` (indices into pair_off) as one batch: -> dict(bases uint8 flat, lens int32, bc_pair_off int64, do_rfa uint8)."""
    lens = np.asarray(lens, dtype=np.int32)
    flat = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
    boff = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)])
    parts, lparts, po = [], [], [0]
    for b in barcodes:
        p0, p1 = int(pair_off[b]), int(pair_off[b + 1])
        parts.append(flat[boff[2 * p0]:boff[2 * p1]])
        lparts.append(lens[2 * p0:2 * p1])
        po.append(po[-1] + (p1 - p0))
    return dict(bases=np.concatenate(parts) if parts else np.zeros(0, np.uint8), lens=np.concatenate(lparts) if lparts else np.zeros(0, np.int32),
                bc_pair_off=np.array(po, dtype=np.int64), do_rfa=np.asarray([do_rfa[b] for b in barcodes], dtype=np.uint8))


_FIELDS_IN = (("bases", np.uint8), ("lens", np.int32), ("bc_pair_off", np.int64), ("do_rfa", np.uint8))
_FIELDS_OUT = (("reg_off", np.int32), ("regs", api.REG_DTYPE), ("alns", api.ALN_DTYPE), ("cigars", np.uint32), ("cand_off", np.int32), ("cands", api.CAND_DTYPE))


def _to_bytes(d, fields):
    """-> (header int64[len(fields)], payload uint8): the arrays back to back, each padded to 8 bytes"""
    hdr = np.array([len(d[k]) for k, _ in fields], dtype=np.int64)
    chunks = []
    for k, dt in fields:
        raw = np.ascontiguousarray(d[k], dtype=dt).view(np.uint8).reshape(-1)
        pad = (-len(raw)) % 8
        chunks.append(raw if not pad else np.concatenate([raw, np.zeros(pad, np.uint8)]))
    return hdr, (np.concatenate(chunks) if chunks else np.zeros(0, np.uint8))


def _from_bytes(hdr, payload, fields):
    out, o = {}, 0
    for (k, dt), n in zip(fields, hdr.tolist()):
        nb = int(n) * np.dtype(dt).itemsize
        out[k] = payload[o:o + nb].view(dt).copy()
        o += nb + ((-nb) % 8)
    return out


class Exchange:
    """Point-to-point byte transport over torch.distributed; tensors live on the GPU for nccl (RCCL), on the host for gloo."""

    def __init__(self, dist, device):
        import torch
        self.dist, self.torch, self.device = dist, torch, device

    def _t(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device != "cpu" else t

    def send(self, dst, hdr, payload):
        self.dist.send(self._t(np.concatenate([[len(payload)], hdr]).astype(np.int64)), dst)
        if len(payload):
            self.dist.send(self._t(payload), dst)

    def recv(self, src, n_fields):
        h = self.torch.zeros(n_fields + 1, dtype=self.torch.int64, device=self.device)
        self.dist.recv(h, src)
        h = h.cpu().numpy()
        p = self.torch.zeros(int(h[0]), dtype=self.torch.uint8, device=self.device)
        if int(h[0]):
            self.dist.recv(p, src)
        return h[1:], p.cpu().numpy()


def scatter_batches(xch, rank, world, packed_per_rank):
    """Rank 0 sends rank r its packed batch (packed_per_rank[r]); every rank returns its own."""
    if rank == 0:
        for r in range(1, world):
            xch.send(r, *_to_bytes(packed_per_rank[r], _FIELDS_IN))
        return packed_per_rank[0]
    hdr, payload = xch.recv(0, len(_FIELDS_IN))
    return _from_bytes(hdr, payload, _FIELDS_IN)


def gather_results(xch, rank, world, mine):
    """Every rank's result slabs (reg_off, regs, alns, cigars, cand_off, cands of its batch) at rank 0, by rank; others get None."""
    if rank != 0:
        xch.send(0, *_to_bytes(mine, _FIELDS_OUT))
        return None
    out = [mine]
    for r in range(1, world):
        hdr, payload = xch.recv(r, len(_FIELDS_OUT))
        out.append(_from_bytes(hdr, payload, _FIELDS_OUT))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Device-resident form of the same dataflow (round 3): a packed batch travels as ONE byte tensor in the memory of the backend's device
# (GPU memory for nccl = RCCL over xGMI; host memory for gloo), the receiving rank hands the library pointers into it
# (arx_batch_reset_device: device-to-device copies, no host hop) and sends its result slabs from where the library left them
# (arx_batch_device_view) -- the host sees a rank's reads and results only at the ingest rank.  All transfers of a phase are posted together
# (torch.distributed.batch_isend_irecv) and the ingest rank runs its own batch while its sends are in flight.
# ---------------------------------------------------------------------------------------------------------------------------------
def _field_offsets(hdr, fields):
    offs, o = [], 0
    for (k, dt), n in zip(fields, [int(x) for x in hdr]):
        nb = n * np.dtype(dt).itemsize
        offs.append((o, nb))
        o += nb + ((-nb) % 8)
    return offs, o


class _DevArray:
    """a device pointer as something torch.as_tensor understands (the CUDA array interface; torch on ROCm speaks it for HIP memory)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = dict(shape=(int(nbytes),), typestr="|u1", data=(int(ptr), False), version=2, strides=None)


def _wrap_bytes(xch, ptr, nbytes):
    """nbytes at `ptr` (memory of the exchange's device) as a uint8 tensor, no copy"""
    torch = xch.torch
    if nbytes == 0 or not ptr:
        return torch.zeros(0, dtype=torch.uint8, device=xch.device)
    if xch.device == "cpu":
        import ctypes
        return torch.from_numpy(np.frombuffer((ctypes.c_uint8 * int(nbytes)).from_address(int(ptr)), dtype=np.uint8))
    return torch.as_tensor(_DevArray(ptr, nbytes), device=xch.device)


def step_device(xch, rank, world, ref, packed_per_rank, handle=None):
    """One scatter -> run -> gather step with device-resident payloads.  packed_per_rank: at rank 0 the packed batches (pack()) of all ranks,
    elsewhere None.  -> (per-rank result slabs as host arrays at rank 0, None elsewhere; this rank's batch handle for reuse)."""
    torch, dist = xch.torch, xch.dist
    P2P = dist.P2POp
    nf_in, nf_out = len(_FIELDS_IN), len(_FIELDS_OUT)

    def wait(works):
        for w in works:
            w.wait()

    # ---- scatter: headers, then payloads; rank 0 posts every peer's transfer at once
    pending, keep = [], []
    if rank == 0:
        hdrs, pays = [], []
        for r in range(1, world):
            h, p = _to_bytes(packed_per_rank[r], _FIELDS_IN)
            hdrs.append(xch._t(np.concatenate([[len(p)], h]).astype(np.int64)))
            pays.append(xch._t(p))                      # the one host -> device copy of this rank's reads, at the ingest rank
        # two groups, as the receivers post them: every peer's header, then every peer's payload (a group is matched as a whole)
        pending = dist.batch_isend_irecv([P2P(dist.isend, hdrs[r - 1], r) for r in range(1, world)]) if world > 1 else []
        ops = [P2P(dist.isend, pays[r - 1], r) for r in range(1, world) if len(pays[r - 1])]
        pending += dist.batch_isend_irecv(ops) if ops else []
        keep = [hdrs, pays]
        mine = packed_per_rank[0]
        hdr_in = np.array([len(mine[k]) for k, _ in _FIELDS_IN], dtype=np.int64)
        pay_in = xch._t(_to_bytes(mine, _FIELDS_IN)[1])
    else:
        h = torch.zeros(nf_in + 1, dtype=torch.int64, device=xch.device)
        wait(dist.batch_isend_irecv([P2P(dist.irecv, h, 0)]))
        hh = h.cpu().numpy()
        pay_in = torch.zeros(int(hh[0]), dtype=torch.uint8, device=xch.device)
        if int(hh[0]):
            wait(dist.batch_isend_irecv([P2P(dist.irecv, pay_in, 0)]))
        hdr_in = hh[1:]
    # ---- run: pointers into the payload go straight to the library
    offs, _tot = _field_offsets(hdr_in, _FIELDS_IN)
    n_reads = int(hdr_in[1])
    slabs = None
    if n_reads > 0:
        if xch.device != "cpu":
            torch.cuda.current_stream().synchronize()   # the payload is complete before another stream reads it
        base = pay_in.data_ptr()
        bco = pay_in[offs[2][0]:offs[2][0] + offs[2][1]].cpu().numpy().view(np.int64)       # barcode offsets and flags: host arguments of arx_batch_rfa, a few KB
        flags = pay_in[offs[3][0]:offs[3][0] + offs[3][1]].cpu().numpy()
        if handle is None:
            handle = ref.batch(np.zeros((2, 0), np.uint8), np.zeros(2, np.int32))
        handle.reset_device(n_reads, int(hdr_in[0]), base + offs[0][0], base + offs[1][0])
        handle.run()
        handle.rfa(bco, flags, fetch=False)
        view = handle.device_view()
        slabs = {k: _wrap_bytes(xch, view[k][0], view[k][1] * view[k][2].itemsize) for k, _ in _FIELDS_OUT}
        counts = np.array([view[k][1] for k, _ in _FIELDS_OUT], dtype=np.int64)
    else:
        counts = np.array([1, 0, 0, 0, 1, 0], dtype=np.int64)
        slabs = {k: torch.zeros(4 if k in ("reg_off", "cand_off") else 0, dtype=torch.uint8, device=xch.device) for k, _ in _FIELDS_OUT}
    wait(pending)
    del keep
    # ---- gather: every rank's counts, then its six slabs, all posted together at rank 0
    if rank != 0:
        wait(dist.batch_isend_irecv([P2P(dist.isend, xch._t(counts), 0)]))
        ops = [P2P(dist.isend, slabs[k], 0) for k, _ in _FIELDS_OUT if len(slabs[k])]
        if ops:
            wait(dist.batch_isend_irecv(ops))
        return None, handle
    cnt = [torch.zeros(nf_out, dtype=torch.int64, device=xch.device) for _ in range(1, world)]
    if world > 1:
        wait(dist.batch_isend_irecv([P2P(dist.irecv, cnt[r - 1], r) for r in range(1, world)]))
    got, ops = [], []
    for r in range(1, world):
        c = cnt[r - 1].cpu().numpy()
        t = {k: torch.zeros(int(n) * np.dtype(dt).itemsize, dtype=torch.uint8, device=xch.device) for (k, dt), n in zip(_FIELDS_OUT, c)}
        ops += [P2P(dist.irecv, t[k], r) for k, _ in _FIELDS_OUT if len(t[k])]
        got.append(t)
    if ops:
        wait(dist.batch_isend_irecv(ops))
    out = []
    for t in [slabs] + got:                             # one device -> host copy per slab, at the ingest rank, where the host consumer is
        out.append({k: t[k].cpu().numpy().view(dt).copy() for k, dt in _FIELDS_OUT})
    return out, handle


def run_batch(ref, packed, handle=None):
    """One packed batch through the whole path on this rank's GPU -> result slabs (host arrays); handle: a Batch to reuse."""
    if len(packed["lens"]) == 0:
        return dict(reg_off=np.zeros(1, np.int32), regs=np.zeros(0, api.REG_DTYPE), alns=np.zeros(0, api.ALN_DTYPE), cigars=np.zeros(0, np.uint32),
                    cand_off=np.zeros(1, np.int32), cands=np.zeros(0, api.CAND_DTYPE)), handle
    b = handle.reset(packed["bases"], packed["lens"]) if handle is not None else ref.batch(packed["bases"], packed["lens"])
    res = b.run().fetch()
    c = b.rfa(packed["bc_pair_off"], packed["do_rfa"])
    return dict(reg_off=res["reg_off"], regs=res["regs"], alns=res["alns"], cigars=res["cigars"], cand_off=c["cand_off"], cands=c["cands"]), b


def merge_in_read_set_order(per_rank, assign, pair_off):
    """The gathered slabs renumbered into the order of the read set: what one batch over all barcodes returns.
    per_rank[r]: slabs of rank r, whose batch holds the barcodes assign[r] in that order; pair_off: pair offsets of the read set."""
    n_bc = len(pair_off) - 1
    n_reads = 2 * int(pair_off[-1])
    where = {}                                          # barcode -> (rank, first read inside the rank's batch)
    for r, bcs in enumerate(assign):
        pos = 0
        for b in bcs:
            where[int(b)] = (r, pos)
            pos += 2 * int(pair_off[b + 1] - pair_off[b])
    reg_off = np.zeros(n_reads + 1, dtype=np.int64)
    cand_off = np.zeros(n_reads + 1, dtype=np.int64)
    regs, alns, cigs, cands = [], [], [], []
    n_reg = n_cig = n_cand = 0
    off64 = [(S["reg_off"].astype(np.int64), S["cand_off"].astype(np.int64)) for S in per_rank]   # once per rank, not per barcode
    for b in range(n_bc):
        r, lr0 = where[b]
        S = per_rank[r]
        g0, nr = 2 * int(pair_off[b]), 2 * int(pair_off[b + 1] - pair_off[b])
        ro, co = off64[r]
        r0, r1 = int(ro[lr0]), int(ro[lr0 + nr])
        c0, c1 = int(co[lr0]), int(co[lr0 + nr])
        reg_off[g0:g0 + nr] = ro[lr0:lr0 + nr] - r0 + n_reg
        cand_off[g0:g0 + nr] = co[lr0:lr0 + nr] - c0 + n_cand
        a = S["alns"][r0:r1].copy()
        w0 = int(a["cigar_off"][0]) if r1 > r0 else 0
        w1 = int(a["cigar_off"][-1] + a["n_cigar"][-1]) if r1 > r0 else 0
        a["cigar_off"] += n_cig - w0
        cd = S["cands"][c0:c1].copy()
        has = cd["reg"] >= 0
        cd["reg"][has] += n_reg - r0                    # region index inside the batch -> inside the read set
        cd["read"] += g0 - lr0
        regs.append(S["regs"][r0:r1]); alns.append(a); cigs.append(S["cigars"][w0:w1]); cands.append(cd)
        n_reg += r1 - r0; n_cig += w1 - w0; n_cand += c1 - c0
    reg_off[n_reads], cand_off[n_reads] = n_reg, n_cand
    if max(n_reg, n_cig, n_cand) >= 1 << 31:
        raise ValueError("the gathered set holds %d regions / %d CIGAR words / %d candidates: beyond the 32-bit offsets of one result slab -- gather fewer barcodes per step" % (n_reg, n_cig, n_cand))
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return dict(reg_off=reg_off.astype(np.int32), regs=cat(regs, api.REG_DTYPE), alns=cat(alns, api.ALN_DTYPE), cigars=cat(cigs, np.uint32),
                cand_off=cand_off.astype(np.int32), cands=cat(cands, api.CAND_DTYPE))
