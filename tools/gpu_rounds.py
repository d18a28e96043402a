#!/usr/bin/env python3
"""Per-launch timing of one device batch (diagnostics): ARX_LAUNCH_LOG lines -> per-kernel histogram by round."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
log = os.path.join(ROOT, "gpurun_out", "launch_log.tsv")
os.makedirs(os.path.dirname(log), exist_ok=True)
if os.path.exists(log):
    os.remove(log)
os.environ["ARX_LAUNCH_LOG"] = log
import numpy as np
import bench
from arachne_amd import api, synth
n_bc = int(sys.argv[1]) if len(sys.argv) > 1 else 350
wname = sys.argv[2] if len(sys.argv) > 2 else "chr20"
wl = bench.WORKLOADS[wname]
prefix = bench.prepare_index("/tmp/arx_bench_cache", wname, wl["lens"], wl["seed"], wl["families"], 0, lambda: None, {})
g = bench.load_genome(prefix)
rs = synth.make_reads(wl["seed"] + 1000, g, n_bc, wl["ppb"], molecules_per_barcode=wl["molecules"], fast=n_bc * wl["ppb"] > 1_500_000)
ref = api.load_reference(prefix)
t_up = time.time()
b = ref.batch(rs.seqs, rs.lens)
t_up = time.time() - t_up
po = rs.pair_offsets()
flags = np.ones(len(po) - 1, dtype=np.uint8)
b.run(api.STAGE_ALN); b.rfa(po, flags, fetch=False)
ref.kernel_times_reset(True)
t = time.time(); b.run(api.STAGE_ALN); t1 = time.time(); b.rfa(po, flags, fetch=False); t2 = time.time()
kt = ref.kernel_times()
print("wall aln %.1f ms, rfa %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
t3 = time.time(); res = b.fetch(); t4 = time.time(); cands = b.rfa(po, flags); t5 = time.time()
nbytes = sum(v.nbytes for v in res.values() if hasattr(v, "nbytes")) + cands["cands"].nbytes + cands["cand_off"].nbytes
print("PCIe side: upload %.1f ms (%d pairs), fetch regions/alignments/CIGARs %.1f ms, rfa again + fetch candidates %.1f ms, %.1f MB of results" % (
    t_up * 1e3, rs.n_pairs, (t4 - t3) * 1e3, (t5 - t4) * 1e3, nbytes / 1e6))
print({k: round(v["ms"], 2) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])})
if os.environ.get("ROUNDS_BRIEF"):
    sys.exit(0)
rows = [r for r in (l.rstrip("\n").split("\t") for l in open(log)) if len(r) == 3]
by = collections.defaultdict(list)
for nm, items, ms in rows:
    by[nm].append((int(items), float(ms)))
for nm in ("extend", "ext_step", "sw_u8", "rescue_step", "reg2aln_nw", "chain", "dedup", "mapq", "rfa", "scan", "seed_bwd", "seed_bwd_wave", "seed_fwd"):
    v = by.get(nm, [])
    print(nm, "launches", len(v), "total ms %.2f" % sum(m for _, m in v))
    for i, (it, ms) in enumerate(v[:12]):
        print("   #%d items %d  %.3f ms  (%.1f ns/item)" % (i, it, ms, 1e6 * ms / max(it, 1)))
    tail = v[12:]
    if tail:
        print("   tail: %d launches, items %d, %.2f ms" % (len(tail), sum(i for i, _ in tail), sum(m for _, m in tail)))
