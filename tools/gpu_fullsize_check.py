#!/usr/bin/env python3
"""Full-size property check on the GPU (run under gpurun): the benchmark workload of BASELINE.json configs[1] (1,000 barcodes x 1,000
pairs on the chr20-size genome) through the whole path under two batch shapes -- three concurrent 350k-pair batches on three streams,
as bench.py runs it, and one batch after the other at 500k pairs -- must give bit-identical regions, alignment records, CIGARs,
placed candidates, MAPQs, mismatch locations, duplicate flags and split records for every read; plus what holds at any size: exactly
one active candidate per read, an active pair is either proper on both mates or on neither, every CIGAR consumes its whole read.
Usage: gpu_fullsize_check.py [barcodes] [pairs_per_barcode] [workload]   (also imported by tests/test_config_shapes.py)"""
import hashlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from arachne_amd import api, synth



def fullsize_check(workload="chr20", n_bc=0, ppb=0, cache="/tmp/arx_bench_cache"):
    """-> (ok, digests, stats, (ref, read set, index prefix)); the caller closes ref"""
    wl = bench.WORKLOADS[workload]
    n_bc, ppb = n_bc or wl["barcodes"], ppb or wl["ppb"]
    prefix = bench.prepare_index(cache, workload, wl["lens"], wl["seed"], wl["families"], 0, lambda: None, {}, alt_spec=wl.get("alt_spec"), decoy_spec=wl.get("decoy_spec"))
    genome = bench.load_genome(prefix)
    rs = bench.workload_reads(wl, wl["seed"] + 1000, genome, n_bc, ppb, fast_above=500_000)
    del genome
    ref = api.load_reference(prefix, 0)
    po = rs.pair_offsets()
    flags_all = np.array([api.worth_running_rfa(rs.barcodes[i], int(po[i + 1] - po[i])) for i in range(len(po) - 1)], dtype=np.uint8)

    def chunks(chunk_pairs):
        start = 0
        while start < len(po) - 1:
            end = start + 1
            while end < len(po) - 1 and po[end + 1] - po[start] <= chunk_pairs:
                end += 1
            yield start, end
            start = end


    def one(start, end):
        p0, p1 = int(po[start]), int(po[end])
        b = ref.batch(rs.seqs[2 * p0:2 * p1], rs.lens[2 * p0:2 * p1]).run()
        out = b.fetch()
        c = b.rfa(po[start:end + 1] - po[start], flags_all[start:end])
        p = b.post()
        b.free()
        return out, c, p


    def digest(parts):
        """per-read content, independent of how the reads were cut into batches"""
        h = {k: hashlib.sha256() for k in ("regs", "alns", "cigars", "cands", "post", "mm_ref", "mm_read", "split")}
        stats = dict(reads=0, regs=0, cands=0, active=0, proper_mismatch=0, bad_cigar=0, dups=0, splits=0)
        for out, c, p in parts:
            h["regs"].update(out["regs"].tobytes())
            a = out["alns"].copy(); a["cigar_off"] = 0
            h["alns"].update(a.tobytes())
            h["cigars"].update(out["cigars"][:int(sum(out["alns"]["n_cigar"]))].tobytes())
            cd = c["cands"].copy(); cd["reg"] = np.where(cd["reg"] >= 0, 0, -1); cd["read"] = 0
            split_rel = p["split"].copy()
            has = split_rel["split"] >= 0
            split_rel["split"][has] -= c["cand_off"][:-1][has]          # candidate index relative to its read
            h["cands"].update(cd.tobytes())
            pp = p["post"].copy(); pp["mm_off"] = 0
            h["post"].update(pp.tobytes()); h["mm_ref"].update(p["mm_ref"].tobytes()); h["mm_read"].update(p["mm_read"].tobytes())
            h["split"].update(split_rel.tobytes())
            act = c["cands"][c["cands"]["active"] == 1]
            stats["reads"] += len(c["cand_off"]) - 1; stats["regs"] += len(out["regs"]); stats["cands"] += len(c["cands"]); stats["active"] += len(act)
            stats["proper_mismatch"] += int((act["is_proper"][0::2] != act["is_proper"][1::2]).sum()) if len(act) == len(c["cand_off"]) - 1 else -1
            al, cg = out["alns"], out["cigars"]
            # query-consuming CIGAR length (M, I, S) per region equals the read length
            ops = np.repeat(np.arange(len(al)), al["n_cigar"])
            words = cg[:len(ops)]
            qlen = np.bincount(ops, weights=np.where(np.isin(words & 15, (0, 1, 3)), words >> 4, 0), minlength=len(al)).astype(np.int64)
            per_read = np.repeat(np.arange(len(out["reg_off"]) - 1), np.diff(out["reg_off"]))
            stats["bad_cigar"] += int((qlen != out["_lens"][per_read]).sum())
            stats["dups"] += int(p["post"]["duplicate"].sum()); stats["splits"] += int(has.sum())
        return {k: v.hexdigest()[:16] for k, v in h.items()}, stats


    def run(chunk_pairs, streams):
        t = time.time()
        cs = list(chunks(chunk_pairs))
        with ThreadPoolExecutor(max_workers=streams) as ex:
            parts = list(ex.map(lambda se: one(*se), cs))
        for (s, e), part in zip(cs, parts):
            part[0]["_lens"] = np.asarray(rs.lens[2 * int(po[s]):2 * int(po[e])])
        d, st = digest(parts)
        print(f"chunk {chunk_pairs} x {streams} streams: {len(cs)} batches, {time.time() - t:.1f}s incl. fetches; {st}", flush=True)
        return d, st


    big = int(os.environ.get("ARX_FULLSIZE_CHUNK", 350_000))
    d1, s1 = run(big, 3)
    d2, s2 = run(big * 10 // 7, 1)
    print("digests A:", d1, flush=True)
    print("digests B:", d2, flush=True)
    ok = d1 == d2 and s1 == s2 and s1["active"] == s1["reads"] and s1["proper_mismatch"] == 0 and s1["bad_cigar"] == 0
    print("FULLSIZE CHECK", "OK" if ok else "FAILED", flush=True)
    return ok, d1, s1, (ref, rs, prefix)


if __name__ == "__main__":
    ok = fullsize_check(sys.argv[3] if len(sys.argv) > 3 else "chr20", int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)[0]
    sys.exit(0 if ok else 1)
