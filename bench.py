#!/usr/bin/env python3
"""bench.py -- paired reads/s through the MI355X hot path (BASELINE.json metric), one process per GPU.

A "step" = one pass of the whole per-barcode path (seed -> locate -> chain -> extend -> rescue -> CIGAR -> candidate
statistics -> RFA joint placement -> MAPQ, i.e. what the reference does per barcode in GetChains + GetAlignments +
tagBestAlignments .. estimateMapQualities, aligner.go:450-490) over the workload of BASELINE.json configs[1]:
a chr20-sized synthetic genome (64,444,167 bp) and 1,000 barcodes x 1,000 pairs of 2x150 bp haplotagging-style reads.
Reads are uploaded to HBM before the timed region; results stay in HBM (PCIe-inclusive numbers: DESIGN.md).
Inside a step the read set is cut into device batches of whole barcodes, each on its own HIP stream and host thread, so
that one batch's latency-bound seeding shares the chip with the others' DP kernels.

N > 1 (launched by torch.distributed.run): barcode groups are independent, so every rank aligns its own barcodes on
its own replica of the index with no data-path collective (weak scaling: each rank gets a full configs[1] read set
with its own seed); torch.distributed (RCCL) is used for the barriers and the max-over-ranks time only.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CHR20_LEN = 64_444_167
SEED0 = 20250905 + 2             # seed = 20250905 + config#


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def prepare_index(cache, genome_len, rank, barrier):
    """Synthetic genome + index (built by the product's own `bwa index` equivalent), cached under /tmp."""
    from arachne_amd import api, synth
    prefix = os.path.join(cache, f"g{genome_len}.fa")
    done = prefix + ".done"
    if rank == 0 and not os.path.exists(done):
        os.makedirs(cache, exist_ok=True)
        t = time.time()
        # chr20-like: one big contig plus two small ones so that contig clamping is exercised
        lens = [genome_len - 2_000_000, 1_500_000, 500_000] if genome_len > 8_000_000 else [genome_len]
        g = synth.make_genome(SEED0, lens)
        g.write_fasta(prefix)
        log(f"genome {genome_len} bp written in {time.time() - t:.1f}s")
        t = time.time()
        api.index_build(prefix, prefix)
        log(f"index built in {time.time() - t:.1f}s")
        np.save(prefix + ".lens.npy", np.array(lens, dtype=np.int64))
        open(done, "w").write("ok")
    barrier()
    return prefix


def load_genome(prefix):
    """Re-read the cached FASTA into the synth.Genome shape make_reads() needs."""
    from arachne_amd import synth
    lens = np.load(prefix + ".lens.npy")
    raw = np.fromfile(prefix, dtype=np.uint8)
    seqs, names, pos = [], [], 0
    lut = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"ACGTN"):
        lut[c] = i
    for k, L in enumerate(lens):
        nl = raw[pos:].tobytes().index(b"\n")
        names.append(raw[pos + 1:pos + nl].tobytes().decode())
        pos += nl + 1
        nlines = (int(L) + 79) // 80
        body = raw[pos:pos + int(L) + nlines]
        s = lut[body]
        seqs.append(s[s != 255])
        assert len(seqs[-1]) == L, (len(seqs[-1]), L)
        pos += int(L) + nlines
    return synth.Genome(names, seqs, [False] * len(seqs))


def cpu_baseline(prefix, rs, n_sample, cores, l_pac, ann_off):
    """Reference C core (oracle/_ref, kind "reference") if the prebuilt .so travelled, else our CPU restatement ("port")."""
    import refdrv
    n = min(n_sample, rs.n_pairs)
    seqs, lens = rs.seqs[:2 * n], rs.lens[:2 * n]
    if refdrv.available():
        r = refdrv.Ref(prefix)
        kind = "reference"
    else:
        import subprocess
        import oradrv
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        r = oradrv.Oracle(prefix)
        kind = "port"
    out = r.batch(seqs, lens, n_threads=cores)
    secs = out["secs"]
    res = dict(value=n / secs, unit="paired reads/s", cores=cores, kind=kind,
               sample=f"first {n} pairs of the same read set, candidate generation + rescue + CIGAR for every candidate "
                      f"(gobwa.go:226-337,400-415 call sequence: the C half of the path), {cores} OpenMP threads, {secs:.1f}s")
    # the Go half (candidate statistics, RFA placement, MAPQ) has no runnable reference: our single-threaded C
    # restatement of it on the whole barcodes of the sample is timed as information only, never folded into `value`
    try:
        import rfadrv
        po = rs.pair_offsets()
        nb = int(np.searchsorted(po, n, side="right")) - 1
        if nb > 0:
            npairs = int(po[nb])
            sub = refdrv_slice(out, 2 * npairs)
            t = time.time()
            rfadrv.oracle_rfa(sub, lens[:2 * npairs], po[:nb + 1], [True] * nb, l_pac, ann_off)
            res["go_half_port_secs"] = time.time() - t
            res["go_half_port_note"] = f"oracle/arx_oracle_rfa.c on {nb} barcodes ({npairs} pairs), 1 thread"
    except Exception as e:
        res["go_half_port_note"] = "not timed: " + repr(e)
    return res, out


def algorithmic_bytes(prefix, rs, n_sample):
    """per-read algorithmic bytes of seeding, counted by the instrumented CPU restatement."""
    import subprocess
    import oradrv
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    o = oradrv.Oracle(prefix)
    n = min(n_sample, rs.n_pairs)
    o.counters(reset=True)
    o.batch(rs.seqs[:2 * n], rs.lens[:2 * n], n_threads=os.cpu_count() or 1)
    c = o.counters()
    reads = 2 * n
    L = float(np.mean(rs.lens[:2 * n]))
    # the seeding kernel runs the first two passes of mem_collect_intv (SMEM search, re-seeding), the third pass has its own
    e1, e2 = c["ext_same_block"] - c["ext3_same_block"], c["ext_two_block"] - c["ext3_two_block"]
    per_read_seed = 64.0 * (e1 + 2 * e2) / reads + L / 4
    per_read_bwd = 64.0 * (c["extb_same_block"] + 2 * c["extb_two_block"]) / reads + L / 4     # backward sweeps (re-read the bases)
    per_read_fwd = 64.0 * ((e1 - c["extb_same_block"]) + 2 * (e2 - c["extb_two_block"])) / reads + L / 4
    per_read_strat = 64.0 * (c["ext3_same_block"] + 2 * c["ext3_two_block"]) / reads + L / 4
    per_read_locate = (64.0 * c["sa_lf_steps"] + 8.0 * c["sa_lookups"]) / reads
    return dict(seed=per_read_seed, fwd=per_read_fwd, bwd=per_read_bwd, strat=per_read_strat, locate=per_read_locate, counters={k: v / reads for k, v in c.items() if k != "n_reads"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--barcodes", type=int, default=1000)
    ap.add_argument("--pairs-per-barcode", type=int, default=1000)
    ap.add_argument("--genome-len", type=int, default=CHR20_LEN)
    ap.add_argument("--chunk-pairs", type=int, default=350_000, help="pairs per device batch inside one step")
    ap.add_argument("--streams", type=int, default=3, help="device batches in flight (one HIP stream + host thread each)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rfa", action="store_true", help="(diagnostics) stop after CIGAR generation")
    ap.add_argument("--post", action="store_true", help="also run the passes between placement and the BAM records in the step (CIGAR walk with "
                    "mismatch locations, markDuplicates, split reads: SURVEY.md s8f-3)")
    ap.add_argument("--depth", type=int, default=1, help="batch handles per chunk of the read set (steps in flight)")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="(diagnostics) all batches start together instead of one seeding stage after the other (about 5 %% more throughput, "
                         "but three seeding stages then compete for HBM latency at once and the roofline kernel's time doubles)")
    ap.add_argument("--cache", default="/tmp/arx_bench_cache")
    ap.add_argument("--lib", default=None, help="(dry runs of this script only) alternative library exporting the C ABI")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for the CPU dry run of the sharding logic)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

        def barrier():
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()
    else:
        def barrier():
            pass

    from arachne_amd import api, synth
    prefix = prepare_index(args.cache, args.genome_len, rank, barrier)
    genome = load_genome(prefix)
    t = time.time()
    rs = synth.make_reads(SEED0 + 1000 * rank, genome, args.barcodes, args.pairs_per_barcode)
    log(f"rank {rank}: {rs.n_pairs} pairs synthesised in {time.time() - t:.1f}s")

    if args.lib:
        api.LIB_PATH = args.lib
        ref = api.Reference(prefix, local_rank, lib_path=args.lib)
    else:
        ref = api.load_reference(prefix, device=local_rank)
        assert ref.backend == "hip:gfx950", ref.backend
    # whole barcodes per device batch; reads go to HBM before the clock starts
    po = rs.pair_offsets()
    # `depth` handles per chunk of the read set: step s works on set s % depth, so that the first batches of a step can start while
    # the last ones of the step before are still in their late stages (each handle has its own stream and work memory)
    sets = []
    for _d in range(max(1, args.depth)):
        bs, start = [], 0
        while start < len(po) - 1:
            end = start + 1
            while end < len(po) - 1 and po[end + 1] - po[start] <= args.chunk_pairs:
                end += 1
            p0, p1 = int(po[start]), int(po[end])
            b = ref.batch(rs.seqs[2 * p0:2 * p1], rs.lens[2 * p0:2 * p1])
            b.bc_pair_off = (po[start:end + 1] - po[start]).astype(np.int64)
            b.do_rfa = np.array([api.worth_running_rfa(rs.barcodes[i], int(po[i + 1] - po[i])) for i in range(start, end)], dtype=np.uint8)
            bs.append(b)
            start = end
        sets.append(bs)
    batches = sets[0]
    log(f"rank {rank}: {len(batches)} device batches uploaded (x{len(sets)} handles)")

    # every device batch has its own HIP stream; host threads drive them concurrently so that the step/DP round trips
    # of one batch overlap the kernels of the others (ctypes releases the GIL inside arx_batch_run)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=len(batches) * len(sets))   # one host thread per batch handle (the staggered schedule needs them all alive)

    import threading

    def run_steps(n_steps):
        """n_steps passes over the whole read set.  Every batch runs start to end on its own stream and host thread; the seeding
        stages go one after the other (batch i of a step after batch i - 1, the first batch of the next step after the last of
        this one), so that the latency-bound seeding kernels share the chip with the VALU-bound DP kernels of the batches ahead
        of them rather than with each other.  Their HIP-event times in the timed region are still co-running times; the same
        kernels alone are measured after the timed region and reported as roofline.isolated."""
        nb = len(batches)
        seeded = [[threading.Event() for _ in range(nb)] for _ in range(n_steps)]

        gate = threading.Semaphore(max(1, args.streams))   # --no-stagger: at most `streams` batches in flight
        failed = threading.Event()

        def wait_for(ev):
            while not ev.wait(0.5):
                if failed.is_set():
                    raise RuntimeError("another batch failed")

        def worker(di):
            try:
                work(di // nb, di % nb)
            except BaseException:
                failed.set()
                raise

        def work(d, i):
            b = sets[d][i]
            for s_ in range(d, n_steps, len(sets)):
                if not args.stagger:
                    with gate:
                        b.run(api.STAGE_ALN)
                        if not args.no_rfa:
                            b.rfa(b.bc_pair_off, b.do_rfa, fetch=False)
                            if args.post:
                                b.post(fetch=False)
                    continue
                if i > 0:
                    wait_for(seeded[s_][i - 1])
                elif s_ > 0:
                    wait_for(seeded[s_ - 1][nb - 1])
                b.run(api.STAGE_SEED)
                seeded[s_][i].set()
                b.run(api.STAGE_ALN)
                if not args.no_rfa:
                    b.rfa(b.bc_pair_off, b.do_rfa, fetch=False)
                    if args.post:
                        b.post(fetch=False)
        list(pool.map(worker, range(nb * len(sets))))

    run_steps(args.warmup)
    ref.kernel_times_reset(True)   # HIP events around every launch on the launch stream, resolved after the timed region
    barrier()
    t0 = time.time()
    run_steps(args.steps)
    barrier()
    dt = time.time() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ktimes = ref.kernel_times()
    counts = [b.counts() for b in batches]
    # the seeding kernel alone on the chip, same inputs, outside the timed region
    ref.kernel_times_reset(True)
    for b in batches:
        b.run(api.STAGE_SEED)
    ktimes_iso = ref.kernel_times()
    ref.kernel_times_reset(False)
    # what every rank did (control plane only; the data path has no collective): pairs and regions per step
    mine = dict(rank=rank, pairs=int(rs.n_pairs), regs=int(sum(c["n_regs"] for c in counts)), read_seed=SEED0 + 1000 * rank)
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        pairs_per_step = sum(r["pairs"] for r in per_rank)
        value = pairs_per_step * args.steps / dt
        baseline = json.load(open(os.path.join(ROOT, "BASELINE.json"))) if os.path.exists(os.path.join(ROOT, "BASELINE.json")) else {}
        out = dict(metric=baseline.get("metric", "paired reads/sec at 1/2/4/8 MI355X, GRCh38 2x150bp; CIGAR/MAPQ match vs CPU"),
                   metric_detail="read pairs per second through the whole per-barcode path (seed + extend + rescue + CIGAR%s), results left in HBM" % ("" if args.no_rfa else " + RFA placement + MAPQ" + (" + CIGAR walk/markDuplicates/split reads" if args.post else "")),
                   value=value, unit="paired reads/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1000.0 * dt / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="u8/i16/i32 integer (max-plus DP) + u64 (FM-index)", data="synthetic",
                   config=dict(workload="BASELINE.json configs[1]: GRCh38 chr20-size genome (%d bp synthetic, planted repeats), "
                                        "%d barcodes x %d pairs 2x150bp per GPU" % (args.genome_len, args.barcodes, args.pairs_per_barcode),
                               pairs_per_step_per_gpu=rs.n_pairs, device_batches=len(batches), parallelism=f"barcode-sharded x{world}"))
        out["per_rank"] = per_rank
        # roofline of the seeding kernel (the HBM-bound headline)
        try:
            ab = algorithmic_bytes(prefix, rs, 10_000)
            reads_per_launch = 2.0 * rs.n_pairs / len(batches)
            # The roofline kernel: the backward sweeps of bwt_smem1a, the largest of the seeding kernels (k_seed_bwd finishes its
            # longest sweeps in k_seed_bwd_wave: the pair is one launch here; each batch launches it for the first pass and for
            # the re-seeding pass, with half of the batch's backward bytes on average).
            def group(kt, names):
                ms = sum(kt[n]["ms"] for n in names if n in kt)
                calls = kt[names[0]]["calls"] if names[0] in kt else 0
                return ms, calls
            ms, calls = group(ktimes, ["seed_bwd", "seed_bwd_wave"])
            if calls:
                launches_per_batch = 2.0
                avg_ms = ms / calls
                bytes_per_launch = ab["bwd"] * reads_per_launch / launches_per_batch
                achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
                # memory-side bytes of the same kernels from the committed rocprofv3 --pmc FETCH_SIZE pass of this command (counters
                # cannot be read from inside the process); null when no such pass is committed
                traffic, traffic_src = None, None
                tf = os.path.join(ROOT, "profiles", "r01", "seed_traffic.json")
                if os.path.exists(tf):
                    tj = json.load(open(tf))
                    if "bwd_fabric_bytes_per_read" in tj:
                        traffic = tj["bwd_fabric_bytes_per_read"] * reads_per_launch / launches_per_batch
                        traffic_src = "profiles/r01/seed_traffic.json"
                out["roofline"] = dict(kernel="seed_bwd (k_seed_bwd + k_seed_bwd_wave: backward sweeps of bwt_smem1a, bwt_extend/bwt_2occ4)", bound="hbm",
                                       achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                                       traffic_unit="bytes per launch", traffic_source=traffic_src, algorithmic_bytes_per_read=ab["bwd"],
                                       reads_per_launch=reads_per_launch, launches_per_batch=launches_per_batch, avg_launch_ms=avg_ms)
                ims, icalls = group(ktimes_iso, ["seed_bwd", "seed_bwd_wave"])
                if icalls:
                    iso_ms = ims / icalls
                    iso = bytes_per_launch / (iso_ms * 1e-3) / 1e9
                    out["roofline"]["isolated"] = dict(achieved=iso, frac=iso / HBM_PEAK_GBS, avg_launch_ms=iso_ms,
                                                       note="same kernels, same inputs, launched alone after the timed region (in the timed region they co-run with the other batches' DP kernels)")
            # the whole seeding stage (forward extensions, backward sweeps, third pass, gathers) against its algorithmic bytes
            stage = ["seed_fwd", "seed_bwd", "seed_bwd_wave", "seed_gather", "seed_strat", "seed_merge"]
            for key, kt in (("roofline_seeding_stage", ktimes), ("roofline_seeding_stage_isolated", ktimes_iso)):
                sms = sum(kt[n]["ms"] for n in stage if n in kt)
                runs = kt["seed_strat"]["calls"] if "seed_strat" in kt else 0
                if runs:
                    tot = (ab["seed"] + ab["strat"]) * reads_per_launch
                    ach = tot / (sms / runs * 1e-3) / 1e9
                    out[key] = dict(kernels=stage, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                    algorithmic_bytes_per_read=ab["seed"] + ab["strat"], ms_per_batch=sms / runs)
            ms, calls = group(ktimes, ["seed_fwd"])
            if calls:
                avg_ms = ms / calls
                ach = ab["fwd"] * reads_per_launch / 2.0 / (avg_ms * 1e-3) / 1e9
                out["roofline_fwd"] = dict(kernel="seed_fwd (k_seed_fwd1 / k_seed_fwd2: forward extensions of bwt_smem1a)", bound="hbm", achieved=ach, peak=HBM_PEAK_GBS,
                                           unit="GB/s", frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_read=ab["fwd"], avg_launch_ms=avg_ms)
            ks3 = ktimes.get("seed_strat")
            if ks3 and ks3["calls"]:
                avg_ms = ks3["ms"] / ks3["calls"]
                ach = ab["strat"] * reads_per_launch / (avg_ms * 1e-3) / 1e9
                out["roofline_strat"] = dict(kernel="seed_strat (k_strat_dyn: bwt_seed_strategy1, third seeding pass)", bound="hbm", achieved=ach, peak=HBM_PEAK_GBS,
                                             unit="GB/s", frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_read=ab["strat"], avg_launch_ms=avg_ms)
            kl = ktimes.get("locate")
            if kl and kl["calls"]:
                avg_ms = kl["ms"] / kl["calls"]
                ach = ab["locate"] * reads_per_launch / (avg_ms * 1e-3) / 1e9
                out["roofline_locate"] = dict(kernel="locate (k_locate_dyn: bwt_sa LF walk)", bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                                              frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_read=ab["locate"], avg_launch_ms=avg_ms,
                                              note="algorithmic bytes are those of the reference's walk to a sample every 32nd row; the device walks to the sample "
                                                   "every %s-th row it builds at arx_open and moves about a quarter of them" % os.environ.get("ARX_SA_DENSE", "8"))
            out["work_per_read"] = ab["counters"]
        except Exception as e:  # the roofline needs the oracle library; never fail the throughput line over it
            log("roofline skipped:", repr(e))
        tot = sum(v["ms"] for v in ktimes.values()) or 1.0
        out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_share"] = {k: round(v["ms"] / tot, 4) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms"])}
        out["rounds"] = dict(ext=max(c["ext_rounds"] for c in counts), rescue=max(c["rescue_rounds"] for c in counts),
                             ext_dp_per_pair=sum(c["n_ext"] for c in counts) / rs.n_pairs, sw_per_pair=sum(c["n_sw"] for c in counts) / rs.n_pairs,
                             regs_per_read=sum(c["n_regs"] for c in counts) / (2.0 * rs.n_pairs))
        if world == 1 and not args.no_cpu_baseline:
            try:
                cores = os.cpu_count() or 1
                n_sample = args.cpu_sample or min(rs.n_pairs, 4000 * cores)
                _n, offs, _l, _a, l_pac = ref.contigs()
                cb, ref_out = cpu_baseline(prefix, rs, n_sample, cores, l_pac, offs)
                out["cpu_baseline"] = cb
                # the same sample through the GPU path must agree with the CPU path it is timed beside
                import parity
                n = min(n_sample, 2000)
                dev = ref.mem_mate_sw(rs.seqs[:2 * n], rs.lens[:2 * n])
                sub = refdrv_slice(ref_out, 2 * n)
                parity.check_final(dev, sub)
                out["parity_checked_pairs"] = n
            except Exception as e:
                log("cpu baseline skipped:", repr(e))
        print(json.dumps(out), flush=True)
    for bs in sets:
        for b in bs:
            b.free()
    ref.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def refdrv_slice(out, n_reads):
    """First n_reads reads of a refdrv/oradrv batch() result."""
    off = out["reg_off"][:n_reads + 1]
    nreg = int(off[-1])
    alns = out["alns"][:nreg]
    ncig = int(alns[-1, 8] + alns[-1, 7]) if nreg else 0
    return dict(reg_off=off, regs=out["regs"][:nreg], alns=alns, cigars=out["cigars"][:ncig])


if __name__ == "__main__":
    main()
