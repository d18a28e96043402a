#!/usr/bin/env python3
"""Generates tests/golden/index_big_v1.json: SHA-256 digests of the index files the REFERENCE's own builder
(bwa_idx_build of /root/reference/src/gobwa/bwa, compiled in place as oracle/_ref/libbwaref.so) writes for a seeded synthetic
genome whose doubled text is just above 2^31 symbols -- the size where 32-bit suffix sorters stop.

Run in the container that holds /root/reference (about half an hour, one core, < 8 GB):
    python tests/golden/make_index_golden.py
The GPU test tests/test_index_build_gpu.py regenerates the same genome from the seed, builds the index with the device builder
of libarachne_amd.so and compares the digests.  Only digests are committed (the files are 2 GB).
"""
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 20250905 + 31
LENS = [1_000_000_000, 73_741_950, 1_234]      # 2 * sum = 2_147_486_368 > 2^31


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def make_genome():
    from arachne_amd import synth
    return synth.make_genome(SEED, LENS)


def main():
    import refdrv
    assert refdrv.available(), "oracle/_ref/libbwaref.so missing: make -C oracle ref"
    d = tempfile.mkdtemp(prefix="arx_goldidx_", dir=os.environ.get("ARX_GOLD_TMP", "/tmp"))
    fa = os.path.join(d, "g.fa")
    t = time.time()
    g = make_genome()
    g.write_fasta(fa)
    print(f"genome written in {time.time() - t:.0f}s", flush=True)
    t = time.time()
    rc = refdrv.Ref().index_build(fa, fa)
    assert rc == 0, rc
    print(f"reference bwa_idx_build: {time.time() - t:.0f}s", flush=True)
    out = dict(seed=SEED, lens=LENS, generator="tests/golden/make_index_golden.py via oracle/_ref (reference bwa_idx_build)",
               fasta_sha256=sha(fa), files={ext: dict(sha256=sha(fa + "." + ext), bytes=os.path.getsize(fa + "." + ext)) for ext in ("bwt", "sa", "pac", "ann", "amb")})
    with open(os.path.join(ROOT, "tests", "golden", "index_big_v1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    for ext in ("", ".bwt", ".sa", ".pac", ".ann", ".amb"):
        os.remove(fa + ext)
    os.rmdir(d)


if __name__ == "__main__":
    main()
