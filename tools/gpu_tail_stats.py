#!/usr/bin/env python3
"""GPU-box diagnostics: distribution of regions / seed occurrences per read on one batch of a bench workload (where the
thread-per-read kernels' tails come from)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from arachne_amd import api, synth
wname = sys.argv[1] if len(sys.argv) > 1 else "grch38"
n_bc = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
wl = bench.WORKLOADS[wname]
prefix = bench.prepare_index("/tmp/arx_bench_cache", wname, wl["lens"], wl["seed"], wl["families"], 0, lambda: None, {})
g = bench.load_genome(prefix)
rs = synth.make_reads(wl["seed"] + 1000, g, n_bc, wl["ppb"], molecules_per_barcode=wl["molecules"])
ref = api.load_reference(prefix)
b = ref.batch(rs.seqs, rs.lens).run()
res = b.fetch()
off, nch, ch, sd = b.debug_chains()
nreg = np.diff(res["reg_off"])
nocc = np.diff(off)
def hist(name, v):
    qs = [50, 90, 99, 99.9, 99.99, 100]
    print(name, "mean %.2f" % v.mean(), {q: int(np.percentile(v, q)) for q in qs}, "share of total in top 0.1%%: %.2f" % (np.sort(v)[-max(1, len(v) // 1000):].sum() / max(1, v.sum())))
hist("regions per read", nreg)
hist("seed occurrences per read", nocc)
hist("chains per read", nch)
pair = nreg[0::2] * nreg[1::2]
hist("regs(R1) x regs(R2) per pair", pair)
print("reads with >= 32 regions:", int((nreg >= 32).sum()), "of", len(nreg), "; >= 100:", int((nreg >= 100).sum()))
