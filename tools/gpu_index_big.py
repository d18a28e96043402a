#!/usr/bin/env python3
"""GPU-box tool: build the index of a GRCh38-size synthetic genome with the device builder and open it (timings of every step).
    python tools/gpu_index_big.py --len 3100000000 [--dir /dev/shm/arx_big]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from arachne_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--len", type=int, default=3_100_000_000)
    ap.add_argument("--dir", default="/dev/shm/arx_big")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    fa = os.path.join(a.dir, f"g{a.len}.fa")
    t = time.time()
    lens = [a.len - 2_000_000, 1_500_000, 500_000]
    g = synth.make_genome(20250905 + 3, lens, fast=True)
    print(f"genome synthesised {time.time() - t:.1f}s", flush=True)
    t = time.time()
    g.write_fasta(fa)
    print(f"fasta written {time.time() - t:.1f}s", flush=True)
    del g
    os.environ["ARX_INDEX_VERBOSE"] = "1"
    t = time.time()
    api.index_build(fa, fa)
    print(f"arx_index_build total {time.time() - t:.1f}s", flush=True)
    for ext in ("bwt", "sa", "pac"):
        print(ext, os.path.getsize(fa + "." + ext), flush=True)
    t = time.time()
    ref = api.load_reference(fa, device=0)
    print(f"arx_open {time.time() - t:.1f}s", flush=True)
    names, offs, clens, alt, l_pac = ref.contigs()
    print("l_pac", l_pac, names, flush=True)
    ref.close()
    if not a.keep:
        for ext in ("", ".bwt", ".sa", ".pac", ".ann", ".amb"):
            os.remove(fa + ext)


if __name__ == "__main__":
    main()
