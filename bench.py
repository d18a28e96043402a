#!/usr/bin/env python3
"""bench.py -- paired reads/s through the MI355X hot path (BASELINE.json metric), one process per GPU.

A "step" = one pass of the whole per-barcode path (seed -> locate -> chain -> extend -> rescue -> CIGAR -> candidate
statistics -> RFA joint placement -> MAPQ, i.e. what the reference does per barcode in GetChains + GetAlignments +
tagBestAlignments .. estimateMapQualities, aligner.go:450-490) over one slice of the workload the metric is quoted on:

  --workload grch38 (default)  BASELINE.json configs[2] on one GPU: a GRCh38-size synthetic genome (24 contigs with the lengths of
                               chr1..22, X, Y: 3,088,269,832 bp, planted repeat families; FM-index 3.1 GB Occ/BWT + 6.2 GB SA
                               sample in HBM, i.e. far outside the 256 MiB Infinity Cache) and a 1,001,000-pair slice per step of
                               TELLseq-like 2x150 bp reads (13,000 barcodes x 77 pairs = the 310 M pairs / 4 M barcodes of the
                               30x set; three device batches in flight).  The index is built by the product itself (arx_index_build: suffix
                               sorting in HBM) inside this script.
  --workload chr20             BASELINE.json configs[1]: 64,444,167 bp, 1,000 barcodes x 1,000 pairs (round 1's configuration;
                               its 64 MB index is Infinity-Cache resident, so its seeding roofline is labelled as such).

Reads are resident in HBM when the clock starts; results stay in HBM (PCIe-inclusive numbers: DESIGN.md).
Inside a step the read set is cut into device batches of whole barcodes, each on its own HIP stream and host thread, so
that one batch's latency-bound seeding shares the chip with the others' DP kernels.

N > 1 (launched by torch.distributed.run): barcode groups are independent, so every rank aligns its own barcodes on
its own replica of the index with no data-path collective (weak scaling: each rank gets a full slice with its own
seed); torch.distributed (RCCL) is used for the barriers and the max-over-ranks time only.

The same job times the reference's C core on the host cores beside the GPU (cpu_baseline) and checks a sample of the GPU
results against it bit for bit; a mismatch makes the run fail (parity_ok false, exit code 1).

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
INFINITY_CACHE_BYTES = 256 << 20 # MI355X_MICROARCH.md "Infinity Cache": a uniformly read table stays resident up to ~255 MiB
CHR20_LEN = 64_444_167
GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
               135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
               50818468, 156040895, 57227415]      # chr1..22, X, Y of GRCh38: 3,088,269,832 bp (a contig must stay below 2^31, bntann1_t.len)
from arachne_amd.synth import segdup_families  # noqa: E402
SEGDUP_FAMILIES = segdup_families(300, 0.71)
WORKLOADS = {
    # name: contig lengths, barcodes, pairs per barcode, molecules per barcode, seed (20250905 + config#), label
    # repeat families at scale 1 are SURVEY.md s8d's recipe as written: 10^4 x 300 bp Alu-like at 12 %, 10^3 x 6 kb L1-like at 5 %,
    # 200 x 50 kb segmental duplications at 1 % (None: make_genome's own scaling, round 1's chr20 genome)
    "grch38": dict(lens=GRCH38_LENS, barcodes=13000, ppb=77, molecules=4, seed=20250905 + 3,
                   families=[(10000, 300, 0.12), (1000, 6000, 0.05), (200, 50000, 0.01)],
                   label="BASELINE.json configs[2] on one GPU: GRCh38-size genome (%d bp synthetic, 24 contigs, planted repeats), TELLseq-like "
                         "%d barcodes x %d pairs 2x150bp per step per GPU (slice of the 30x set)"),
    "chr20": dict(lens=[CHR20_LEN - 2_000_000, 1_500_000, 500_000], barcodes=1000, ppb=1000, molecules=10, seed=20250905 + 2, families=None,
                  label="BASELINE.json configs[1]: GRCh38 chr20-size genome (%d bp synthetic, planted repeats), %d barcodes x %d pairs 2x150bp per GPU"),
    # configs[3]: GRCh38 + ALT/decoy contigs (.alt-flagged diverged copies of primary slices, unflagged decoys), repeat-enriched stLFR-like set:
    # half of the molecules drawn from the planted families (weights: the low-copy segmental duplications real genomes have many of 0.71,
    # L1-like 0.15, Alu-like 0.1, the 200-copy family 0.04 -- on a scaled-down genome the restatement counts 8 regions per read, median 2,
    # 99th percentile 107, and 8 rescue alignments per pair, against 1.7 and 0.17 on the TELLseq-like default), 70 % of their pairs
    # overlapping the copy; stLFR-like barcodes hold few pairs
    "alt_repeat": dict(lens=GRCH38_LENS, barcodes=33000, ppb=30, molecules=2, seed=20250905 + 4,
                       families=[(10000, 300, 0.12, 0.1), (1000, 6000, 0.05, 0.15), (200, 50000, 0.01, 0.04)] + SEGDUP_FAMILIES,
                       alt_spec=(220, 100_000, 900_000, 0.01), decoy_spec=(400, 15_000), reads=dict(repeat_bias=0.5, barcode_style="stlfr"), chunk_pairs=350_000, depth=1,
                       label="BASELINE.json configs[3]: GRCh38-size genome + ALT/decoy contigs (%d bp synthetic, 220 .alt-flagged ALT contigs, 400 decoys), "
                             "repeat-enriched stLFR-like set (half of the molecules from planted repeat families / segmental duplications), %d barcodes x %d pairs 2x150bp per step per GPU"),
    # configs[4]: configs[1] with 30 % of the pairs VX:i:0 in dash-less barcodes of 1-4 pairs (worthRunningRFA false, aligner.go:469-477,1018-1030)
    "vxmix": dict(lens=[CHR20_LEN - 2_000_000, 1_500_000, 500_000], barcodes=1000, ppb=1000, molecules=10, seed=20250905 + 5, families=None,
                  reads=dict(invalid_frac=0.3),
                  label="BASELINE.json configs[4]: chr20-size genome (%d bp synthetic), %d barcodes x %d pairs 2x150bp per GPU of which 30 %% VX:i:0 in "
                        "dash-less barcodes of 1-4 pairs (single-read fallback) filed among the RFA barcodes"),
}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def prepare_index(cache, name, lens, seed, families, rank, barrier, setup, in_child=False, alt_spec=None, decoy_spec=None):
    """Synthetic genome + index (built by the product's own `bwa index` equivalent: suffix sorting in HBM), cached on disk.
    The work is done by a child process (bench.py --prepare-only): a process that has synthesised a 3 GB genome and sorted 6.2 G
    suffixes through 125 GB of HBM runs the timed region ~25 % slower afterwards (host-side round trips of the stage loops take longer;
    measured in round 2) -- the measuring process stays clean."""
    from arachne_amd import api, synth
    total = int(sum(lens))
    prefix = os.path.join(cache, f"{name}_{total}.fa")
    done = prefix + ".done"
    if rank == 0 and not os.path.exists(done) and not in_child:
        import subprocess
        os.makedirs(cache, exist_ok=True)
        spec = prefix + ".spec.json"
        json.dump(dict(cache=cache, name=name, lens=[int(x) for x in lens], seed=int(seed), families=families, alt_spec=alt_spec, decoy_spec=decoy_spec), open(spec, "w"))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--prepare-only", spec])
        setup.update(json.load(open(prefix + ".setup.json")))
    if rank == 0 and not os.path.exists(done):
        os.makedirs(cache, exist_ok=True)
        t = time.time()
        g = synth.make_genome(seed, lens, repeat_families=families, fast=total > 500_000_000, alt_spec=alt_spec, decoy_spec=decoy_spec)
        setup["genome_synth_s"] = round(time.time() - t, 2)
        t = time.time()
        g.write_fasta(prefix)
        if any(g.alt):
            g.write_alt(prefix + ".alt")                 # read by arx_open like bns_restore reads it (bntseq.c:98-206)
        np.save(prefix + ".codes.npy", np.concatenate(g.seqs))
        np.save(prefix + ".lens.npy", np.array([len(x) for x in g.seqs], dtype=np.int64))
        np.save(prefix + ".copies.npy", g.copies)
        json.dump(dict(names=g.names, alt=[bool(a) for a in g.alt], fam_weight=[float(x) for x in g.fam_weight]), open(prefix + ".contigs.json", "w"))
        del g
        setup["genome_write_s"] = round(time.time() - t, 2)
        log(f"genome {total} bp synthesised and written in {setup['genome_synth_s'] + setup['genome_write_s']:.1f}s")
        t = time.time()
        api.index_build(prefix, prefix)
        setup["index_build_s"] = round(time.time() - t, 2)
        log(f"index built in {time.time() - t:.1f}s (arx_index_build: FASTA -> .pac/.ann/.amb on the host, BWT + SA in HBM)")
        if total > 500_000_000:
            os.remove(prefix)          # the codes are kept as .npy; the FASTA of a GRCh38-size genome is 3 GB
        json.dump(setup, open(prefix + ".setup.json", "w"))
        open(done, "w").write("ok")
    barrier()
    return prefix


def load_genome(prefix):
    """The cached genome in the synth.Genome shape make_reads() needs (contigs are views of one memory-mapped array)."""
    from arachne_amd import synth
    lens = np.load(prefix + ".lens.npy")
    cat = np.load(prefix + ".codes.npy", mmap_mode="r")
    off = np.concatenate([[0], np.cumsum(lens)])
    seqs = [cat[int(off[i]):int(off[i + 1])] for i in range(len(lens))]
    meta = json.load(open(prefix + ".contigs.json"))
    return synth.Genome(meta["names"], seqs, meta["alt"], np.load(prefix + ".copies.npy"), np.array(meta["fam_weight"]))


def workload_reads(wl, seed, genome, n_barcodes, ppb, fast_above=1_500_000):
    """The read set of a workload entry (WORKLOADS[...]['reads'] holds what differs from the TELLseq-like default)."""
    from arachne_amd import synth
    return synth.make_reads(seed, genome, n_barcodes, ppb, molecules_per_barcode=wl["molecules"], fast=n_barcodes * ppb > fast_above, **wl.get("reads", {}))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(prefix, rs, n_sample, cores, l_pac, ann_off):
    """Reference C core (oracle/_ref, kind "reference") if the prebuilt .so travelled, else our CPU restatement ("port").
    -> (cpu_baseline dict, the reference's results for the sample, the driver object)"""
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
    res = dict(value=n / secs, unit="paired reads/s", cores=cores, kind=kind, cpu_model=cpu_model(), nproc=os.cpu_count(),
               sample=f"first {n} pairs of the same read set, candidate generation + rescue + CIGAR for every candidate "
                      f"(gobwa.go:226-337,400-415 call sequence: the C half of the path), {cores} OpenMP threads, {secs:.1f}s")
    # the reference's default is -t 8 (main.go:40): the same code on 8 threads, on a smaller sample
    n8 = min(n, 24_000)
    o8 = r.batch(seqs[:2 * n8], lens[:2 * n8], n_threads=8)
    res["t8"] = dict(value=n8 / o8["secs"], cores=8, sample=f"first {n8} pairs, {o8['secs']:.1f}s")
    if kind == "reference":   # thread-seconds per phase (seed / extend / rescue / CIGAR), timing-only replay of the same calls
        ns = min(n, 250 * cores)
        ph = r.phase_split(seqs[:2 * ns], lens[:2 * ns], n_threads=cores)
        tot = ph["seed"] + ph["extend"] + ph["rescue"] + ph["cigar"]
        res["phase_share"] = {k: round(ph[k] / tot, 4) for k in ("seed", "extend", "rescue", "cigar")}
        res["phase_note"] = (f"thread-seconds on {ns} pairs, {cores} threads: seed = mem_chain .. mem_flt_chained_seeds (SMEM, locate, chaining), extend = "
                             "mem_chain2aln + mem_sort_dedup_patch, rescue = mem_matesw loops, cigar = mem_reg2aln")
    return res, out, r


def go_half_port(rs, ref_out, n, l_pac, ann_off):
    """The Go half (candidate statistics, RFA placement, MAPQ) has no runnable reference: our single-threaded C restatement
    of it on the whole barcodes inside the first n pairs -> (timing info, its result, pairs covered, barcode offsets)."""
    import rfadrv
    po = rs.pair_offsets()
    nb = int(np.searchsorted(po, n, side="right")) - 1
    if nb <= 0:
        return None
    npairs = int(po[nb])
    sub = refdrv_slice(ref_out, 2 * npairs)
    from arachne_amd import api
    flags = [api.worth_running_rfa(rs.barcodes[i], int(po[i + 1] - po[i])) for i in range(nb)]
    t = time.time()
    orfa = rfadrv.oracle_rfa(sub, rs.lens[:2 * npairs], po[:nb + 1], flags, l_pac, ann_off)
    secs = time.time() - t
    return dict(secs=secs, note=f"oracle/arx_oracle_rfa.c on {nb} barcodes ({npairs} pairs), 1 thread"), orfa, npairs, po[:nb + 1], flags


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
    per_read_locate = (64.0 * c["sa_lf_steps"] + 8.0 * c["sa_lookups"]) / reads       # the reference's walk: to a sample every 32nd row
    per_read_locate8 = (64.0 * c["sa_lf_steps8"] + 8.0 * c["sa_lookups"]) / reads     # the same lookups walked to a sample every 8th row
    per_read_locate1 = 8.0 * c["sa_lookups"] / reads                                       # ... with the whole array resident: the entries themselves
    per_read_locate4 = (64.0 * c.get("sa_lf_steps4", c["sa_lf_steps8"] * 3.0 / 7.0) + 8.0 * c["sa_lookups"]) / reads   # ... every 4th row (the product's default since round 3)
    o.close()
    return dict(seed=per_read_seed, fwd=per_read_fwd, bwd=per_read_bwd, strat=per_read_strat, locate=per_read_locate, locate8=per_read_locate8, locate4=per_read_locate4, locate1=per_read_locate1,
                counters={k: v / reads for k, v in c.items() if k != "n_reads"})


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--prepare-only":   # the child of prepare_index()
        sp = json.load(open(sys.argv[2]))
        fam = [tuple(f) for f in sp["families"]] if sp["families"] is not None else None
        prepare_index(sp["cache"], sp["name"], sp["lens"], sp["seed"], fam, 0, lambda: None, {}, in_child=True, alt_spec=sp.get("alt_spec"), decoy_spec=sp.get("decoy_spec"))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="grch38", choices=sorted(WORKLOADS), help="grch38: the configuration the metric is quoted on (default); chr20: configs[1]; alt_repeat: configs[3]; vxmix: configs[4]")
    ap.add_argument("--barcodes", type=int, default=0, help="override the workload's barcodes per step")
    ap.add_argument("--pairs-per-barcode", type=int, default=0, help="override the workload's pairs per barcode")
    ap.add_argument("--genome-len", type=int, default=0, help="(experiments) one big contig of this length plus two small ones instead of the workload's genome")
    ap.add_argument("--chunk-pairs", type=int, default=0, help="pairs per device batch inside one step (0: the workload's shape -- one batch per step and two steps in "
                    "flight on the TELLseq-like sets, three 350,000-pair batches per step on the repeat-rich one whose work memory per pair is several times larger)")
    ap.add_argument("--streams", type=int, default=3, help="device batches in flight (one HIP stream + host thread each)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rfa", action="store_true", help="(diagnostics) stop after CIGAR generation")
    ap.add_argument("--post", action="store_true", help="also run the passes between placement and the BAM records in the step (CIGAR walk with "
                    "mismatch locations, markDuplicates, split reads: SURVEY.md s8f-3)")
    ap.add_argument("--boundary-steps", type=int, default=12, help="extra steps timed boundary to boundary (host reads in through arx_batch_reset, "
                    "results out through arx_batch_detach + arx_batch_fetch_detached into reused page-locked host arrays); 0: skip.  Reported under `boundary`, never "
                    "as `value`.  Several steps per handle: with one step per handle (3 until the end of round 3) the figure is the latency of one batch, not a rate")
    ap.add_argument("--scatter-steps", type=int, default=0, help="N > 1: extra steps run as SURVEY.md s8e's dataflow -- rank 0 owns every rank's barcodes, assigns whole "
                    "barcodes by pair count (LPT), scatters the packed batches and gathers the result slabs over torch.distributed point-to-point (RCCL / xGMI) "
                    "inside the timed steps; reported under `scatter_gather` (an ingest rank cannot feed 8 GPUs at kernel rate: see DESIGN.md s6)")
    ap.add_argument("--no-end-to-end", dest="end_to_end", action="store_false", help="skip the FASTQ -> BAM pass (reported under `end_to_end`, never as `value`)")
    ap.add_argument("--e2e-workers", type=int, default=6, help="file pairs / worker threads of the end-to-end pass")
    ap.add_argument("--depth", type=int, default=0, help="batch handles per chunk of the read set (steps in flight); 0: the workload's shape")
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
    wl = dict(WORKLOADS[args.workload])
    if args.genome_len:
        wl["lens"] = [args.genome_len - 2_000_000, 1_500_000, 500_000] if args.genome_len > 8_000_000 else [args.genome_len]
    n_barcodes, ppb = args.barcodes or wl["barcodes"], args.pairs_per_barcode or wl["ppb"]
    # Batch shape (round 3, measured on the default command: 350 k x 3 batches per step 8.73 M pairs/s, 520 k x 2 with two steps in flight 8.36 M,
    # ONE 1,001 k batch per step with two steps in flight 8.95-9.13 M, with three in flight 9.73 M, with four 8.94 M -- the one-wavefront tails of
    # a batch (rescue replay, chaining of the few reads in high-copy repeats) are paid once per step instead of three times.  A 1 M-pair batch holds
    # ~37 GB of work memory since the seeding passes hand theirs back to the arena (65 GB before: three in flight beside the 69 GB k-mer table
    # ran at 9.07 M, four did not fit).
    if args.chunk_pairs <= 0:
        args.chunk_pairs = wl.get("chunk_pairs", 1_001_000)
    if args.depth <= 0:
        args.depth = wl.get("depth", 3 if args.chunk_pairs >= n_barcodes * ppb else 1)
    genome_len = int(sum(wl["lens"]))       # primary contigs; ALT / decoy contigs come on top (index_bytes_in_files has the whole index)
    SEED0 = wl["seed"]
    setup = {}
    if args.lib:
        api.LIB_PATH = args.lib
    prefix = prepare_index(args.cache, args.workload if not args.genome_len else "custom", wl["lens"], SEED0, wl["families"], rank, barrier, setup,
                           alt_spec=wl.get("alt_spec"), decoy_spec=wl.get("decoy_spec"))
    genome = load_genome(prefix)
    t = time.time()
    rs = workload_reads(wl, SEED0 + 1000 * (rank + 1), genome, n_barcodes, ppb)
    setup["reads_synth_s"] = round(time.time() - t, 2)
    log(f"rank {rank}: {rs.n_pairs} pairs synthesised in {time.time() - t:.1f}s")
    del genome

    t = time.time()
    if args.lib:
        ref = api.Reference(prefix, local_rank, lib_path=args.lib)
    else:
        ref = api.load_reference(prefix, device=local_rank)
        assert ref.backend == "hip:gfx950", ref.backend
    setup["arx_open_s"] = round(time.time() - t, 2)
    index_info = ref.index_info()           # what arx_open built: k-mer tables, the whole suffix array and its inverse (text mode) or the sample
    index_bytes = {ext: os.path.getsize(prefix + "." + ext) for ext in ("bwt", "sa", "pac")}
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
            b.host = (b.pin(np.array(rs.seqs[2 * p0:2 * p1], copy=True).reshape(-1)), np.ascontiguousarray(rs.lens[2 * p0:2 * p1]))   # the caller's own, page-locked read buffer
            b.out = {}
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
    fetch_pool = ThreadPoolExecutor(max_workers=len(batches) * len(sets))   # the second host thread of every handle (boundary pass: results home while the next reads run)
    phase_lock = threading.Lock()
    phase_s = {False: {}, True: {}}    # host-side seconds per phase of a batch's pass, summed over batches (resident / boundary passes)

    def run_steps(n_steps, boundary=False):
        """n_steps passes over the whole read set (boundary: every pass hands the reads over from host memory and takes the results back).  Every batch runs start to end on its own stream and host thread; the seeding
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
                tp = [time.time()]
                if boundary:
                    b.reset(*b.host)                      # reads come from host memory the caller page-locked (arx_host_register): DMA on the batch's stream
                tp.append(time.time())
                if i > 0:
                    wait_for(seeded[s_][i - 1])
                elif s_ > 0:
                    wait_for(seeded[s_ - 1][nb - 1])
                tp.append(time.time())
                b.run(api.STAGE_SEED)
                seeded[s_][i].set()
                tp.append(time.time())
                b.run(api.STAGE_ALN)
                tp.append(time.time())
                if not args.no_rfa:
                    b.rfa(b.bc_pair_off, b.do_rfa, fetch=False)
                    if args.post:
                        b.post(fetch=False)
                tp.append(time.time())
                if boundary:
                    # regions, alignment records, CIGARs, placed candidates with MAPQ back in host memory: copied aside on the device (arx_batch_detach),
                    # then taken home by a second thread (arx_batch_fetch_detached, its own stream) while this one goes on with the next reads
                    if os.environ.get("ARX_BENCH_SYNC_FETCH"):        # (A/B: the handle's own thread takes them home before it goes on)
                        b.fetch_into(b.out)
                    else:
                        if getattr(b, "pending", None) is not None:
                            b.pending.result()
                        b.pending = fetch_pool.submit(b.fetch_detached_into, b.out, b.detach())
                tp.append(time.time())
                with phase_lock:
                    for k_, (a_, b_) in zip(("reset", "wait_turn", "seed", "align", "rfa", "fetch"), zip(tp, tp[1:])):
                        phase_s[boundary][k_] = phase_s[boundary].get(k_, 0.0) + (b_ - a_)
                    phase_s[boundary]["batches"] = phase_s[boundary].get("batches", 0) + 1
        list(pool.map(worker, range(nb * len(sets))))
        for bs_ in sets:                                      # the last results are home before the clock stops
            for b_ in bs_:
                if getattr(b_, "pending", None) is not None:
                    b_.pending.result(); b_.pending = None

    # every handle's first run obtains its work memory (hipMalloc of tens of GB: seconds beside a 69 GB k-mer table): part of the set-up like
    # the upload of the reads, whatever --warmup says
    t = time.time()
    for bs in sets:
        for b in bs:
            b.run(api.STAGE_ALN)
            if not args.no_rfa:
                b.rfa(b.bc_pair_off, b.do_rfa, fetch=False)
                if args.post:
                    b.post(fetch=False)
    setup["prime_handles_s"] = round(time.time() - t, 2)
    run_steps(args.warmup)
    phase_s[False].clear()
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
    # the same steps boundary to boundary: host arrays in (arx_batch_reset), host arrays out (arx_batch_fetch, arx_batch_rfa_fetch)
    boundary = None
    if args.boundary_steps > 0 and args.stagger:
        ref.kernel_times_reset(False)
        run_steps(1, boundary=True)                       # sizes the reused host arrays and the staging
        phase_s[True].clear()
        barrier()
        tb = time.time()
        run_steps(args.boundary_steps, boundary=True)
        barrier()
        boundary = time.time() - tb
        if dist is not None:
            import torch
            tt = torch.tensor([boundary], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            boundary = float(tt.item())
    # the seeding kernel alone on the chip, same inputs, outside the timed region
    ref.kernel_times_reset(True)
    for b in batches:
        b.run(api.STAGE_SEED)
    ktimes_iso = ref.kernel_times()
    # every kernel with the chip to itself: the same batches through the whole path one after the other (in the timed region three batches'
    # streams overlap, and the elapsed time of a launch there includes what it waits for)
    ref.kernel_times_reset(True)
    for b in batches:
        b.run(api.STAGE_SEED)
        b.run(api.STAGE_ALN)
        if not args.no_rfa:
            b.rfa(b.bc_pair_off, b.do_rfa, fetch=False)
    ktimes_alone = ref.kernel_times()
    ref.kernel_times_reset(False)
    # what every rank did (control plane only; the data path has no collective): pairs and regions per step
    # SURVEY.md s8e as written: one ingest rank, LPT assignment of whole barcodes, scatter of packed batches, gather of result slabs
    scatter_info = None
    if args.scatter_steps > 0 and dist is not None and args.stagger:
        from arachne_amd import shard
        xch = shard.Exchange(dist, "cuda" if args.backend == "nccl" else "cpu")
        nb = len(batches)
        packed = assign = None
        if rank == 0:   # the ingest rank holds every rank's barcodes (the same read sets the ranks synthesised for themselves)
            genome = load_genome(prefix)
            sets_all = [rs] + [workload_reads(wl, SEED0 + 1000 * (r + 1), genome, n_barcodes, ppb) for r in range(1, world)]
            del genome
            seqs_all = np.concatenate([x.seqs for x in sets_all]); lens_all = np.concatenate([x.lens for x in sets_all])
            po_all = np.concatenate([[0]] + [x.pair_offsets()[1:] + i * rs.n_pairs for i, x in enumerate(sets_all)]).astype(np.int64)
            names_all = [nm for x in sets_all for nm in x.barcodes]
            flags_all = np.array([api.worth_running_rfa(names_all[i], int(po_all[i + 1] - po_all[i])) for i in range(len(po_all) - 1)], dtype=np.uint8)
            assign = shard.lpt_assign(np.diff(po_all), world)
            packed = [shard.pack(seqs_all, lens_all, po_all, flags_all, a) for a in assign]
            del sets_all, seqs_all

        state = dict(h=sets[0][0])   # an existing handle takes the received reads (arx_batch_reset_device): no further work memory beside the resident batches

        def scatter_step():
            # device-resident payloads (arachne_amd/shard.py: step_device): every peer's transfer posted together, the library reads the received
            # tensor and hands out its result slabs where they lie; one device batch per rank and step
            gathered, state["h"] = shard.step_device(xch, rank, world, ref, packed, state["h"])
            return gathered

        scatter_step()
        barrier()
        ts = time.time()
        for _ in range(args.scatter_steps):
            scatter_step()
        barrier()
        ts = time.time() - ts
        if rank == 0:
            loads = [int(np.diff(po_all)[a].sum()) for a in assign]
            scatter_info = dict(value=world * rs.n_pairs * args.scatter_steps / ts, unit="paired reads/s", steps=args.scatter_steps, ms_per_step=1000.0 * ts / args.scatter_steps,
                                pairs_per_rank=loads, note="rank 0 owns all barcodes, LPT assignment of whole barcodes by pair count; packed batches travel as device tensors "
                                "(torch.distributed batch_isend_irecv over %s, all peers posted together), the library reads them in place (arx_batch_reset_device) and the result "
                                "slabs are sent from where it left them (arx_batch_device_view); the ingest rank runs its own batch while its sends are in flight and copies every "
                                "rank's slabs to the host once" % args.backend)
    mine = dict(rank=rank, pairs=int(rs.n_pairs), regs=int(sum(c["n_regs"] for c in counts)), read_seed=SEED0 + 1000 * (rank + 1))
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    parity_failed = None
    if rank == 0:
        pairs_per_step = sum(r["pairs"] for r in per_rank)
        value = pairs_per_step * args.steps / dt
        baseline = json.load(open(os.path.join(ROOT, "BASELINE.json"))) if os.path.exists(os.path.join(ROOT, "BASELINE.json")) else {}
        out = dict(metric=baseline.get("metric", "paired reads/sec at 1/2/4/8 MI355X, GRCh38 2x150bp; CIGAR/MAPQ match vs CPU"),
                   metric_detail="read pairs per second through the whole per-barcode path (seed + extend + rescue + CIGAR%s), reads resident in HBM before the clock starts, results left in HBM (transfers excluded)" % ("" if args.no_rfa else " + RFA placement + MAPQ" + (" + CIGAR walk/markDuplicates/split reads" if args.post else "")),
                   value=value, unit="paired reads/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1000.0 * dt / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="u8/i16/i32 integer (max-plus DP) + u64 (FM-index)", data="synthetic",
                   config=dict(workload=wl["label"] % (genome_len, n_barcodes, ppb), genome_bp=genome_len,
                               index_bytes_in_files=index_bytes, pairs_per_step_per_gpu=rs.n_pairs, device_batches=len(batches),
                               parallelism=f"barcode-sharded x{world}"))
        out["per_rank"] = per_rank
        # how often the whole path ran over the read set in this process (profilers sum over all of it): priming of every handle, warm-up, timed
        # steps, the boundary pass (one sizing step + its steps), the "alone" pass
        out["whole_path_passes"] = len(sets) + args.warmup + args.steps + (1 + args.boundary_steps if (args.boundary_steps > 0 and args.stagger) else 0) + 1
        out["setup_s"] = setup
        out["index"] = index_info
        if boundary:
            out_bytes = sum(int(c["n_regs"]) * (88 + 48) + int(c["n_cigar"]) * 4 + int(c["n_reads"]) * 8 for c in counts) + sum(b._n_cands for b in batches) * 96
            if scatter_info:
                out["scatter_gather"] = scatter_info
            out["boundary"] = dict(value=pairs_per_step * args.boundary_steps / boundary, unit="paired reads/s", steps=args.boundary_steps,
                                   ms_per_step=1000.0 * boundary / args.boundary_steps,
                                   host_bytes_in_per_pair=round(float(rs.lens.sum()) / rs.n_pairs + 8, 1), host_bytes_out_per_pair=round(out_bytes / rs.n_pairs, 1),
                                   note="same steps timed from host arrays in (arx_batch_reset from the caller's page-locked read buffer, arx_host_register: DMA on the "
                                        "batch's stream, handle and work memory reused) to host arrays out (arx_batch_detach, then arx_batch_fetch_detached by the handle's second host thread into page-locked "
                                        "arrays the caller reuses, while the handle takes its next reads); PCIe-inclusive, never `value`")
        def per_batch(d):
            n = max(d.get("batches", 0), 1)
            return {k: round(1000.0 * v / n, 1) for k, v in d.items() if k != "batches"}
        out["host_phase_ms_per_batch"] = dict(resident=per_batch(phase_s[False]), boundary=per_batch(phase_s[True]),
                                              note="wall time of a batch's host thread per phase (three batches in flight: the phases of different batches overlap); "
                                                   "wait_turn = waiting for the batch ahead to leave the seeding stage (the staggered schedule)")
        # where the FM-index lives decides what the seeding kernels are bound by: a table under 256 MiB stays in the Infinity Cache
        occ_bytes = index_bytes["bwt"]
        in_hbm = occ_bytes > INFINITY_CACHE_BYTES
        bound = "hbm" if in_hbm else "infinity-cache"
        bound_note = ("Occ/BWT table %.2f GB in HBM, scattered 64-byte block reads" % (occ_bytes / 1e9)) if in_hbm else \
            ("Occ/BWT table %.0f MB fits the 256 MiB Infinity Cache: the fraction below is against the HBM peak for comparison only, it is not an HBM measurement" % (occ_bytes / 1e6))
        # roofline of the seeding kernel (the HBM-bound headline)
        try:
            ab = algorithmic_bytes(prefix, rs, 10_000)
            reads_per_launch = 2.0 * rs.n_pairs / len(batches)
            # The roofline kernel: the backward sweeps of bwt_smem1a, the largest of the seeding kernels (k_seed_bwd finishes its
            # longest sweeps in k_seed_bwd_wave: the pair is one launch here; each batch launches it for the first pass and for
            # the re-seeding pass, with half of the batch's backward bytes on average).
            BWD = ["seed_bwd", "seed_bwd32", "seed_bwd64", "seed_bwd_wave"]   # one "launch" of the backward sweeps = the group kernels of the three list-length bins + the wave kernel

            def group(kt, names):
                ms = sum(kt[n]["ms"] for n in names if n in kt)
                calls = kt[names[0]]["calls"] if names[0] in kt else 0
                return ms, calls
            ms, calls = group(ktimes, BWD)
            if calls:
                launches_per_batch = 2.0
                avg_ms = ms / calls
                bytes_per_launch = ab["bwd"] * reads_per_launch / launches_per_batch
                achieved = bytes_per_launch / max(avg_ms * 1e-3, 1e-12) / 1e9
                # memory-side bytes of the same kernels from the committed rocprofv3 --pmc FETCH_SIZE pass of this command (counters
                # cannot be read from inside the process); null when no such pass is committed
                traffic, traffic_src = None, None
                rel = os.path.join("profiles", "r03", "seed_traffic_%s.json" % args.workload)
                if not os.path.exists(os.path.join(ROOT, rel)):
                    rel = os.path.join("profiles", "r02", "seed_traffic_%s.json" % args.workload)
                tf = os.path.join(ROOT, rel)
                if os.path.exists(tf):
                    tj = json.load(open(tf))
                    if "bwd_fabric_bytes_per_read" in tj:
                        traffic = tj["bwd_fabric_bytes_per_read"] * reads_per_launch / launches_per_batch
                        traffic_src = rel
                out["roofline"] = dict(kernel="seed_bwd (k_seed_bwd_g<16/32/64> + k_seed_bwd_wave: backward sweeps of bwt_smem1a, bwt_extend/bwt_2occ4)", bound=bound, bound_note=bound_note,
                                       achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                                       traffic_unit="bytes per launch", traffic_source=traffic_src, algorithmic_bytes_per_read=ab["bwd"],
                                       reads_per_launch=reads_per_launch, launches_per_batch=launches_per_batch, avg_launch_ms=avg_ms)
                ims, icalls = group(ktimes_iso, BWD)
                if icalls:
                    iso_ms = ims / icalls
                    iso = bytes_per_launch / max(iso_ms * 1e-3, 1e-12) / 1e9
                    out["roofline"]["isolated"] = dict(achieved=iso, frac=iso / HBM_PEAK_GBS, avg_launch_ms=iso_ms,
                                                       note="same kernels, same inputs, launched alone after the timed region (in the timed region they co-run with the other batches' DP kernels)")
            # the whole seeding stage (forward extensions, backward sweeps, third pass, gathers) against its algorithmic bytes
            stage = ["seed_pack", "seed_fwd", "seed_bwd", "seed_bwd32", "seed_bwd64", "seed_bwd_wave", "seed_gather", "seed_strat", "seed_merge"]
            for key, kt in (("roofline_seeding_stage", ktimes), ("roofline_seeding_stage_isolated", ktimes_iso)):
                sms = sum(kt[n]["ms"] for n in stage if n in kt)
                runs = kt["seed_strat"]["calls"] if "seed_strat" in kt else 0
                if runs:
                    tot = (ab["seed"] + ab["strat"]) * reads_per_launch
                    ach = tot / max(sms / runs * 1e-3, 1e-12) / 1e9
                    out[key] = dict(kernels=stage, bound=bound, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                    algorithmic_bytes_per_read=ab["seed"] + ab["strat"], ms_per_batch=sms / runs)
            ms, calls = group(ktimes, ["seed_fwd"])
            if calls:
                avg_ms = ms / calls
                ach = ab["fwd"] * reads_per_launch / 2.0 / max(avg_ms * 1e-3, 1e-12) / 1e9
                out["roofline_fwd"] = dict(kernel="seed_fwd (k_seed_fwd1 / k_seed_fwd2: forward extensions of bwt_smem1a)", bound=bound, achieved=ach, peak=HBM_PEAK_GBS,
                                           unit="GB/s", frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_read=ab["fwd"], avg_launch_ms=avg_ms)
            ks3 = ktimes.get("seed_strat")
            if ks3 and ks3["calls"]:
                avg_ms = ks3["ms"] / ks3["calls"]
                ach = ab["strat"] * reads_per_launch / max(avg_ms * 1e-3, 1e-12) / 1e9
                out["roofline_strat"] = dict(kernel="seed_strat (k_strat_dyn: bwt_seed_strategy1, third seeding pass)", bound=bound, achieved=ach, peak=HBM_PEAK_GBS,
                                             unit="GB/s", frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_read=ab["strat"], avg_launch_ms=avg_ms)
            kl = ktimes.get("locate")
            if kl and kl["calls"]:
                avg_ms = kl["ms"] / kl["calls"]
                # the bytes of the walk the kernel does: the same lookups, each walked to the first row that is a multiple of the
                # device's sample interval (counted by the restatement for 8; the reference's own walk to every 32nd row beside it)
                dense = index_info["sa_rows_per_entry"]
                per_read = ab["locate1"] if dense == 1 else ab["locate4"] if dense == 4 else ab["locate8"] if dense == 8 else ab["locate"]
                ach = per_read * reads_per_launch / max(avg_ms * 1e-3, 1e-12) / 1e9
                out["roofline_locate"] = dict(kernel=("locate (KLocate: one entry of the resident suffix array per occurrence, no walk)" if dense == 1 else
                                                      "locate (k_locate_dyn: bwt_sa LF walk to the suffix-array sample every %d-th row)" % dense), bound=bound,
                                              achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_read=per_read,
                                              reference_walk_bytes_per_read=ab["locate"], avg_launch_ms=avg_ms)
            out["work_per_read"] = ab["counters"]
            cw = ab["counters"]
            # SURVEY.md s8d's path-level figure: pairs/s x algorithmic bytes per pair / HBM peak, bytes per read =
            # 64 (E1 + 2 E2) + (64 S + 8 N_sa) + L/4 + 88 N_reg, all counted by the instrumented restatement on a sample of the same reads
            per_read = 64.0 * (cw["ext_same_block"] + 2 * cw["ext_two_block"]) + 64.0 * cw["sa_lf_steps"] + 8.0 * cw["sa_lookups"] + float(np.mean(rs.lens)) / 4 + 88.0 * cw["n_regs"]
            out["roofline_path"] = dict(definition="SURVEY.md s8d: pairs/s x algorithmic bytes per pair / HBM peak; bytes per read = 64(E1+2E2) + 64 S + 8 N_sa + L/4 + 88 N_reg "
                                                   "(the reference's own walks, incl. its suffix-array sample every 32nd row)",
                                        algorithmic_bytes_per_pair=2 * per_read, achieved=value * 2 * per_read / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=value * 2 * per_read / 1e9 / HBM_PEAK_GBS)
            # Smith-Waterman kernels: cell updates per second (the reference's cells, counted by the restatement: band cells of ksw_extend2,
            # query x window cells of both ksw_u8 passes, band cells of ksw_global2) against the VALU; instruction counters from the committed
            # rocprofv3 --pmc pass of this command (profiles/r03/sw_counters_<workload>.json, tools/prof_summary.py), null when none is committed
            swc = {}
            rel_sw = os.path.join("profiles", "r03", "sw_counters_%s.json" % args.workload)
            if os.path.exists(os.path.join(ROOT, rel_sw)):
                swc = json.load(open(os.path.join(ROOT, rel_sw)))
            sw = {}
            for key, kern, cells_key, what in (("extend", "extend", "cells_extend", "ksw_extend2 band cells (k_extend_g16)"), ("sw_u8", "sw_u8", "cells_u8", "ksw_u8 cells, both passes (k_sw_u8_g16)"),
                                               ("reg2aln_nw", "reg2aln_nw", "cells_global", "ksw_global2 band cells (k_reg2aln_nw_g16)")):
                cells_step = cw[cells_key] * 2.0 * rs.n_pairs
                e = dict(cells=what, cells_per_read=cw[cells_key])
                if kern in ktimes and ktimes[kern]["ms"] > 0:
                    e["gcups_timed"] = cells_step * args.steps / (ktimes[kern]["ms"] * 1e-3) / 1e9
                if kern in ktimes_alone and ktimes_alone[kern]["ms"] > 0:
                    e["gcups_alone"] = cells_step / (ktimes_alone[kern]["ms"] * 1e-3) / 1e9
                e.update(swc.get(kern, dict(valu_busy=None, lane_ops_per_cell=None, lds_insts_per_valu=None)))
                sw[key] = e
            out["roofline_sw"] = dict(kernels=sw, bound="valu", counters_source=rel_sw if swc else None,
                                      peak_note="gfx950: 256 CUs x 4 SIMDs; valu_busy = rocprofv3's VALUBusy (SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles at %.1f GHz); "
                                                "lane_ops_per_cell = SQ_INSTS_VALU x 64 / reference cells; max-plus DP, no MFMA" % 2.4)
            if "roofline" in out:   # what the committed counters say the roofline kernel is bound by
                rc = swc.get("seed_bwd")
                if rc and rc.get("valu_busy") is not None:
                    out["roofline"]["valu_busy"] = rc["valu_busy"]
                    if rc["valu_busy"] >= 0.6 and in_hbm:
                        out["roofline"]["bound"] = "valu"
                        out["roofline"]["bound_note"] += "; the kernel's vector instructions keep the SIMDs busy %.0f %% of its time (VALUBusy, committed --pmc pass): instruction issue, not HBM, bounds it; frac stays the HBM-roofline figure the metric asks for" % (100 * rc["valu_busy"])
        except Exception as e:  # the roofline needs the oracle library; never fail the throughput line over it
            log("roofline skipped:", repr(e))
        tot = sum(v["ms"] for v in ktimes.values()) or 1.0
        out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_share"] = {k: round(v["ms"] / tot, 4) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms"])}
        tot_a = sum(v["ms"] for v in ktimes_alone.values()) or 1.0
        out["kernel_ms_per_step_alone"] = {k: round(v["ms"], 3) for k, v in sorted(ktimes_alone.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_share_alone"] = {k: round(v["ms"] / tot_a, 4) for k, v in sorted(ktimes_alone.items(), key=lambda kv: -kv[1]["ms"])}
        book = ("chain", "chain_heavy", "ext_step", "rescue_step", "rescue_heavy", "dedup", "dedup_heavy", "scan")
        out["bookkeeping_share"] = dict(kernels=list(book), overlapped=round(sum(ktimes.get(k, {"ms": 0.0})["ms"] for k in book) / tot, 4),
                                        alone=round(sum(ktimes_alone.get(k, {"ms": 0.0})["ms"] for k in book) / tot_a, 4),
                                        note="list bookkeeping (chaining, extension state machines, rescue replay, de-duplication, scans) as a share of the "
                                             "summed kernel time; round 1: 0.31 overlapped on the chr20-size workload")
        out["kernel_share_note"] = ("kernel_share: elapsed time per launch (HIP events on the launch stream) summed over the timed region, where the "
                                    "streams of %d device batches overlap -- a launch that shares or waits for CUs counts for as long as it is in flight, "
                                    "so one-wavefront tails (rescue_heavy, chain_heavy) weigh far more than the resources they hold; "
                                    "kernel_share_alone: one step's batches run one after the other after the timed region, every launch alone on the chip"
                                    % len(batches))
        out["rounds"] = dict(ext=max(c["ext_rounds"] for c in counts), rescue=max(c["rescue_rounds"] for c in counts),
                             ext_dp_per_pair=sum(c["n_ext"] for c in counts) / rs.n_pairs, sw_per_pair=sum(c["n_sw"] for c in counts) / rs.n_pairs,
                             regs_per_read=sum(c["n_regs"] for c in counts) / (2.0 * rs.n_pairs))
        out["rccl_ranks"] = world if (dist is not None and args.backend == "nccl") else 0
        # the boundary pass handed the same reads over from host memory (arx_batch_reset) and took the results back (arx_batch_fetch +
        # arx_batch_rfa_fetch into reused arrays): its last step's output of the first batch must equal the resident run's, byte for byte
        if boundary and not args.no_rfa and batches[0].out and not scatter_info:
            b0 = batches[0]
            c0 = b0.counts()
            res_out = b0.fetch()
            res_c = np.zeros(b0._n_cands, dtype=api.CAND_DTYPE); res_off = np.zeros(b0.n_reads + 1, dtype=np.int32)
            ref._check(ref.lib.arx_batch_rfa_fetch(ref.h, b0.h, res_off.ctypes.data, res_c.ctypes.data))
            same = (np.array_equal(b0.out["reg_off"][:b0.n_reads + 1], res_out["reg_off"]) and np.array_equal(b0.out["regs"][:c0["n_regs"]], res_out["regs"])
                    and np.array_equal(b0.out["alns"][:c0["n_regs"]], res_out["alns"]) and np.array_equal(b0.out["cigars"][:c0["n_cigar"]], res_out["cigars"])
                    and np.array_equal(b0.out["cand_off"][:b0.n_reads + 1], res_off) and b0.out["cands"][:b0._n_cands].tobytes() == res_c.tobytes())
            out["boundary"]["matches_resident"] = bool(same)
            if not same:
                parity_failed = "boundary pass (arx_batch_reset / fetch_into) differs from the resident run on the first device batch"
        out["parity_ok"] = None if parity_failed is None else False
        if world != 1:
            out["parity_skipped"] = "N > 1: the CPU leg and the parity gate run on the N = 1 line only"
        elif args.no_cpu_baseline:
            out["parity_skipped"] = "--no-cpu-baseline"
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            n_sample = args.cpu_sample or min(rs.n_pairs, 2000 * cores)
            _n, offs, _l, _a, l_pac = ref.contigs()
            cb = ref_out = None
            try:
                cb, ref_out, _drv = cpu_baseline(prefix, rs, n_sample, cores, l_pac, offs)
                out["cpu_baseline"] = cb
            except (ImportError, OSError, RuntimeError) as e:   # the baseline library is missing or cannot load the index: report, keep the throughput line
                log("cpu baseline skipped:", repr(e))
                out["parity_skipped"] = "the CPU baseline library could not run: " + repr(e)[:200]
            if ref_out is not None:
                # The same sample through the GPU path must agree with the CPU path it is timed beside: regions, positions, strands, NM and
                # CIGARs against the reference's C core, candidates / placement / MAPQ against the restatement of the Go half run on the
                # reference's C-half output.  A mismatch fails the run.
                import parity
                try:
                    n = min(n_sample, 2000)
                    gh = go_half_port(rs, ref_out, n, l_pac, offs)
                    n_chk = gh[2] if gh else n
                    b = ref.batch(rs.seqs[:2 * n_chk], rs.lens[:2 * n_chk]).run()
                    dev = b.fetch()
                    parity.check_final(dev, refdrv_slice(ref_out, 2 * n_chk))
                    out["parity_checked_pairs"] = n_chk
                    if gh and not args.no_rfa:
                        info, orfa, _np, bpo, flags = gh
                        cb["go_half_port_secs"] = info["secs"]
                        cb["go_half_port_note"] = info["note"]
                        parity.check_rfa(b.rfa(bpo, flags), orfa)
                        out["parity_checked_rfa_barcodes"] = len(flags)
                    b.free()
                    out["parity_ok"] = parity_failed is None
                except AssertionError as e:
                    out["parity_ok"] = False
                    out["parity_error"] = str(e)[:500]
                    parity_failed = str(e)
        # End to end (SURVEY.md s8d "reported separately"): the same read set as barcode-sorted FASTQ files on disk -> arx_feeder -> the path ->
        # arx_recbuf (AppendBam's record logic) -> arx_bam (BGZF), the loop of Arachne() (aligner.go:335-371, bamwriter.go:615-658) as
        # arachne_amd/e2e.py drives it through the C ABI; one worker thread per file pair.  Never `value`.
        if args.end_to_end and world == 1 and not args.no_rfa:
            try:
                from arachne_amd import e2e
                import shutil
                ed = os.path.join(args.cache, "e2e_%d" % os.getpid())
                os.makedirs(ed, exist_ok=True)
                t = time.time()
                k = max(1, args.e2e_workers)
                cuts = [int(po[(len(po) - 1) * i // k]) for i in range(k)] + [rs.n_pairs]
                files = []
                for i in range(k):
                    f1, f2 = os.path.join(ed, "r1_%d.fq" % i), os.path.join(ed, "r2_%d.fq" % i)
                    synth.write_fastq_fast(rs, f1, f2, cuts[i], cuts[i + 1])
                    files.append((f1, f2))
                t_w = time.time() - t
                for bs in sets:                      # the resident batches give their memory back to the end-to-end workers' handles
                    for b in bs:
                        b.free()
                sets, batches = [], []
                st = e2e.run(ref, files, os.path.join(ed, "out"), pairs_per_batch=max(20000, rs.n_pairs // (2 * k)), bam_threads=8, rec_threads=8,
                             lib_path=args.lib or api.LIB_PATH, warm_passes=1)
                out["end_to_end"] = dict(value=st["pairs_per_s"], unit="paired reads/s", pairs=st["pairs"], seconds=round(st["seconds"], 3), workers=k,
                                         fastq_bytes=sum(os.path.getsize(f) for pr in files for f in pr), bam_bytes=st.get("bam_bytes"), fastq_write_s=round(t_w, 2),
                                         worker_seconds={kk: round(st[kk], 3) for kk in ("feeder_s", "device_s", "fetch_s", "records_s", "bam_s")},
                                         warm_passes=st.get("warm_passes", 0),
                                         note="steady state: the second pass over the files, through the batch handles the first pass created (a handle's first batch pays for its "
                                              "work memory once); FASTQ files on disk (plain, barcode-sorted, one pair per worker) -> arx_feeder_next -> arx_batch_reset/run/rfa/post -> "
                                              "arx_batch_fetch + rfa_fetch + post_fetch -> arx_recbuf_build (primary record per read) -> arx_bam_write (BGZF level 1, one BAM per "
                                              "worker); wall clock over all workers; worker_seconds are summed over the workers")
                shutil.rmtree(ed, ignore_errors=True)
            except Exception as e:  # the headline number does not depend on this pass
                log("end-to-end pass failed:", repr(e))
                out["end_to_end"] = dict(error=repr(e)[:300])
        print(json.dumps(out), flush=True)
    for bs in sets:
        for b in bs:
            b.free()
    ref.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and parity_failed:
        log("PARITY MISMATCH between the GPU path and the CPU path:", parity_failed)
        sys.exit(1)


def refdrv_slice(out, n_reads):
    """First n_reads reads of a refdrv/oradrv batch() result."""
    off = out["reg_off"][:n_reads + 1]
    nreg = int(off[-1])
    alns = out["alns"][:nreg]
    ncig = int(alns[-1, 8] + alns[-1, 7]) if nreg else 0
    return dict(reg_off=off, regs=out["regs"][:nreg], alns=alns, cigars=out["cigars"][:ncig])


if __name__ == "__main__":
    main()
