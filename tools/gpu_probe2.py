"""Bisect the dedup stage on the GPU: compare raw (pre-dedup) regions GPU vs host test double, then partial dedup modes."""
import os, sys, tempfile, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from arachne_amd import api
import workloads, parity
mode = sys.argv[1]; n = int(sys.argv[2])
os.environ["ARX_DEDUP_DBG"] = mode
z = np.load(os.path.join(ROOT, "tests", "golden", "bwa_path_v1.npz"))
prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_probe_"))
seqs, lens = z["reads"][:n], z["lens"][:n]
out = {}
for name, lib in (("gpu", os.environ.get("ARX_LIB", api.LIB_PATH)), ("sim", os.path.join(ROOT, "tests", "hostsim", "libarx_hostsim.so"))):
    ref = api.Reference(prefix, lib_path=lib)
    b = ref.batch(seqs, lens)
    t = time.time(); b.run(3); dt = time.time() - t
    n_core, rg = b.debug_core(); occ_off = b.debug_chains()[0]
    out[name] = (n_core.copy(), rg.copy(), occ_off.copy())
    print(name, "mode", mode, "stage3 %.3fs" % dt, "sum n_core", int(n_core.sum()), flush=True)
(na, ra, oa), (nb, rb_, ob) = out["gpu"], out["sim"]
assert (oa == ob).all()
bad = 0
for r in range(n):
    if na[r] != nb[r]: bad += 1; print("read", r, "n differs", na[r], nb[r]); continue
    x = parity.regs_to_rows(ra[oa[r]:oa[r] + na[r]]); y = parity.regs_to_rows(rb_[oa[r]:oa[r] + nb[r]])
    if not (x == y).all(): bad += 1; print("read", r, "rows differ\n", x, "\n", y)
print("mode", mode, "reads differing:", bad, flush=True)
