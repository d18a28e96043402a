"""Step-by-step probe of the device path on a GPU box: every stage logs to gpurun_out/probe.log as it goes."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "probe.log"), "a")
T0 = time.time()


def log(*a):
    msg = "[%7.2fs] " % (time.time() - T0) + " ".join(str(x) for x in a)
    print(msg, flush=True)
    LOG.write(msg + "\n")
    LOG.flush()


def main():
    import faulthandler
    faulthandler.enable(file=LOG)
    faulthandler.dump_traceback_later(240, exit=True, file=LOG)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    log("start; n reads", n)
    from arachne_amd import api
    import oradrv
    import parity
    import workloads
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    z = np.load(os.path.join(ROOT, "tests", "golden", "bwa_path_v1.npz"))
    prefix = workloads.unpack_index(z, tempfile.mkdtemp(prefix="arx_probe_"))
    log("index unpacked")
    ref = api.Reference(prefix, 0, lib_path=os.environ.get('ARX_LIB', api.LIB_PATH))
    log("opened", ref.backend)
    o = oradrv.Oracle(prefix)
    seqs, lens = z["reads"][:n], z["lens"][:n]
    ref.kernel_times_reset(True)
    b = ref.batch(seqs, lens)
    log("batch uploaded")
    for stage, name in ((1, "seed"), (2, "chain"), (3, "extend"), (4, "rescue"), (5, "aln")):
        t = time.time()
        b.run(stage)
        log("stage", name, "done in %.3fs" % (time.time() - t), b.counts())
        if stage == 1:
            parity.check_intervals(b, o, seqs, lens)
            log("  intervals match")
        if stage == 2:
            parity.check_chains(b, o, seqs, lens)
            log("  chains match")
        if stage == 3:
            parity.check_core(b, o, seqs, lens)
            log("  core regions match")
    dev = b.fetch()
    parity.check_final(dev, o.batch(seqs, lens))
    log("final match:", len(dev["regs"]), "regions")
    for k, v in sorted(ref.kernel_times().items(), key=lambda kv: -kv[1]["ms"]):
        log("  kernel %-12s %9.3f ms  calls %6d items %9d" % (k, v["ms"], v["calls"], v["items"]))
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
