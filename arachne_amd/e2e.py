"""End to end: FASTQ pairs in, BAM files out -- the loop of the reference's Arachne() (src/aligner/aligner.go:335-371: a producer reading
barcode sets, `-t` workers calling DoRFAForOneBarcode, one BamThread writing, bamwriter.go:615-658) re-shaped around the device path:

    k file pairs, each with its own worker thread:
        arx_feeder_next (whole barcode sets, ~pairs_per_batch pairs)  ->  arx_batch_reset / run / rfa / post on the worker's own stream
        ->  arx_batch_fetch + arx_batch_rfa_fetch + arx_batch_post_fetch into reused host arrays  ->  arx_recbuf_build (AppendBam's record
        logic on host threads)  ->  arx_bam_write (BGZF on host threads) into the worker's own BAM

Everything between the file reads and the file writes goes through the C ABI of include/arachne_amd.h; this module is the host-side
mirror of the Go driver (Python here because the image has no Go toolchain; INTEGRATION.md has the Go form).  The split /
supplementary records and the position-bucketed second copy of every record the reference writes (bamwriter.go:279-281) are not produced.
"""
from __future__ import annotations

import threading
import time

import os

import numpy as np

from . import api

_TRACE = bool(os.environ.get("ARX_E2E_TRACE"))


def run(ref: api.Reference, fastq_pairs, out_prefix: str, pairs_per_batch: int = 250_000, bam_threads: int = 8, rec_threads: int = 8, level: int = 1,
        penalty: float = -4, lib_path: str = api.LIB_PATH, warm_passes: int = 0):
    """fastq_pairs: [(r1, r2), ...] barcode-sorted files (plain or gzip), one worker each.  -> stats dict (pairs, seconds, pairs/s, per-stage
    seconds summed over workers).  warm_passes: untimed passes over the same files first, through the same batch handles -- a handle's first
    batch pays for its work memory (hipMalloc of several GiB: seconds once a 69 GB k-mer table sits beside it), which a run over a whole
    read set pays once; the stats are those of the last pass."""
    names, offs, clens, alt, l_pac = ref.contigs()
    stats = dict(pairs=0, records=0, batches=0, feeder_s=0.0, device_s=0.0, fetch_s=0.0, records_s=0.0, bam_s=0.0)
    lock = threading.Lock()
    errors = []

    gate = threading.Barrier(len(fastq_pairs) + 1)
    t_pass = [0.0] * (warm_passes + 2)

    def worker(k, r1, r2):
        try:
            batch, buf = None, {}
            rb = [api.RecBuf(lib_path=lib_path), api.RecBuf(lib_path=lib_path)]   # two record buffers: one is written out while the next is built
            for ps in range(warm_passes + 1):
                gate.wait()
                batch, buf = one_pass(k, r1, r2, batch, buf, rb, ps == warm_passes)
                gate.wait()
            if batch is not None:
                batch.free()
            for x in rb:
                x.free()
        except BaseException as e:  # noqa: BLE001 -- reported by the caller's thread
            errors.append(e)
            gate.abort()

    def one_pass(k, r1, r2, batch, buf, rb, counted):
        if True:
            fd = api.Feeder(r1, r2, lib_path=lib_path)
            bam = api.BamWriter(f"{out_prefix}.{k}.bam", names, clens, extra_header="@PG\tID:arachne_amd\n", threads=bam_threads, level=level, lib_path=lib_path)
            loc = dict(pairs=0, records=0, batches=0, feeder_s=0.0, device_s=0.0, fetch_s=0.0, records_s=0.0, bam_s=0.0)
            # the worker's BamThread (bamwriter.go:615-658): record views are compressed and written by a thread of their own while the worker
            # goes on with the next barcode sets; a view's record buffer is reused two batches later, when its write has returned
            import queue
            wq = queue.Queue(maxsize=1)
            written = [threading.Event(), threading.Event()]
            for e_ in written:
                e_.set()
            werr = []

            def writer():
                while True:
                    item = wq.get()
                    if item is None:
                        return
                    slot, view = item
                    try:
                        t_ = time.time()
                        bam.write_view(view)
                        loc["bam_s"] += time.time() - t_
                    except BaseException as e:  # noqa: BLE001
                        werr.append(e)
                    finally:
                        written[slot].set()
            wt = threading.Thread(target=writer)
            wt.start()
            while True:
                t0 = time.time()
                nx = fd.next_raw(pairs_per_batch)
                t1 = time.time()
                if nx is None:
                    break
                sb, v = nx
                batch = batch.reset(v["bases"], v["lens"]) if batch is not None else ref.batch(v["bases"], v["lens"])
                batch.run(api.STAGE_ALN)
                batch.rfa(v["set_pair_off"], v["do_rfa"], penalty=penalty, fetch=False)
                t2 = time.time()
                if _TRACE:
                    print(f"[e2e] worker {k} batch {loc['batches']}: {int(v['n_pairs'])} pairs, feeder {t1 - t0:.3f}s device {t2 - t1:.3f}s", flush=True)
                batch.fetch_into(buf)
                post = batch.post_into(buf)
                t3 = time.time()
                slot = loc["batches"] & 1
                written[slot].wait()                    # the buffer's last view is on disk
                if werr:
                    raise werr[0]
                view = rb[slot].build(sb, buf["cand_off"], buf["cands"], buf["alns"], buf["cigars"], post, threads=rec_threads)
                t4 = time.time()
                written[slot].clear()
                wq.put((slot, view))
                loc["pairs"] += int(v["n_pairs"]); loc["records"] += int(view.n_records); loc["batches"] += 1
                loc["feeder_s"] += t1 - t0; loc["device_s"] += t2 - t1; loc["fetch_s"] += t3 - t2; loc["records_s"] += t4 - t3
            wq.put(None)
            wt.join()
            if werr:
                raise werr[0]
            t5 = time.time()
            st = bam.close()
            loc["bam_s"] += time.time() - t5
            fd.close()
            if counted:
                with lock:
                    for key, val in loc.items():
                        stats[key] += val
                    stats.setdefault("bam_bytes", 0)
                    stats["bam_bytes"] += st["bytes_out"]
            return batch, buf

    th = [threading.Thread(target=worker, args=(k, r1, r2)) for k, (r1, r2) in enumerate(fastq_pairs)]
    for x in th:
        x.start()
    t = time.time()
    try:
        for ps in range(warm_passes + 1):
            gate.wait()                 # the pass starts
            t = time.time()
            gate.wait()                 # ... and is over when every worker has closed its BAM
            t_pass[ps] = time.time() - t
    except threading.BrokenBarrierError:
        pass
    for x in th:
        x.join()
    if errors:
        raise errors[0]
    stats["seconds"] = t_pass[warm_passes]
    stats["warm_passes"] = warm_passes
    stats["pairs_per_s"] = stats["pairs"] / stats["seconds"] if stats["seconds"] > 0 else 0.0
    stats["workers"] = len(fastq_pairs)
    return stats
