#!/usr/bin/env python3
"""Where a boundary-to-boundary step spends its time: arx_batch_reset and arx_batch_fetch with pageable and with page-locked caller arrays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from arachne_amd import api, synth
g = synth.make_genome(5, [8_000_000], repeat_families=[(30, 2000, 0.01)], fast=True)
rs = synth.make_reads(6, g, 40, 10000)
import tempfile
d = tempfile.mkdtemp(); fa = os.path.join(d, "g.fa"); g.write_fasta(fa); api.index_build(fa, fa)
ref = api.load_reference(fa, 0)
po = rs.pair_offsets()
flags = [True] * (len(po) - 1)
seqs = np.array(rs.seqs, copy=True).reshape(-1); lens = np.ascontiguousarray(rs.lens)
b = ref.batch(seqs, lens)
def step(buf, label):
    t0 = time.time(); b.reset(seqs_in, lens); t1 = time.time(); b.run(); t2 = time.time(); b.rfa(po, flags, fetch=False); t3 = time.time(); c = b.fetch_into(buf); t4 = time.time()
    nbytes = sum(v.nbytes for k, v in buf.items() if v is not None)
    print("%s: reset %.1f ms  run %.1f  rfa %.1f  fetch %.1f ms (buffers %.0f MB, regs %d)" % (label, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), nbytes / 1e6, c["n_regs"]), flush=True)
seqs_in = seqs
rc = ref.lib.arx_host_register(seqs.ctypes.data, 4096); print("probe register rc", rc); ref.lib.arx_host_unregister(seqs.ctypes.data)
api.Batch.pin_saved = api.Batch.pin
api.Batch.pin = lambda self, a: a            # pageable
buf = {}
for i in range(3): step(buf, "pageable")
api.Batch.pin = api.Batch.pin_saved
seqs_in = b.pin(np.array(seqs, copy=True))
buf = {}
for i in range(3): step(buf, "page-locked")
