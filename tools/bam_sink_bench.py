#!/usr/bin/env python3
"""Throughput of the BAM sink (arx_bam_*) on this host: typical 2x150 bp records (name, one CIGAR op, 150 bases + qualities, ~45 bytes
of aux), one batch of n records written `reps` times, for several thread counts.  Usage: bam_sink_bench.py [n_records] [threads ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
from arachne_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
threads = [int(x) for x in sys.argv[2:]] or [1, 8, 32]
rng = np.random.default_rng(1)
names = [b"r%09d" % i for i in range(n)]
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, 150))
qual = (rng.integers(2, 41, size=(n, 150)) + 33).astype(np.uint8)
aux1 = b"RGZlib1\0ASC\x96XMZ0\0AMZ1\0XTC\0BXZA01C02B03D04-1\0VXC\x01"
lib = api._load(api.LIB_PATH)
name_off = np.arange(n + 1, dtype=np.int64) * 10
name_b = np.frombuffer(b"".join(names), dtype=np.uint8)
flag = np.full(n, 99, np.int32); rid = np.zeros(n, np.int32); pos = rng.integers(0, 200_000_000, size=n).astype(np.int32); mapq = np.full(n, 60, np.uint8)
mrid = np.zeros(n, np.int32); mpos = pos + 200; tlen = np.full(n, 350, np.int32)
cig_off = np.arange(n + 1, dtype=np.int64); cig = np.full(n, 150 << 4, np.uint32)
seq_off = np.arange(n + 1, dtype=np.int64) * 150
aux_off = np.arange(n + 1, dtype=np.int64) * len(aux1); aux_b = np.frombuffer(aux1 * n, dtype=np.uint8)
b = api._BamBatch(n, name_off.ctypes.data, name_b.ctypes.data, flag.ctypes.data, rid.ctypes.data, pos.ctypes.data, mapq.ctypes.data, mrid.ctypes.data, mpos.ctypes.data,
                  tlen.ctypes.data, cig_off.ctypes.data, cig.ctypes.data, seq_off.ctypes.data, seq.ctypes.data, qual.ctypes.data, 33, aux_off.ctypes.data, aux_b.ctypes.data)
out = "/dev/shm/arx_bam_bench.bam" if os.path.isdir("/dev/shm") else "/tmp/arx_bam_bench.bam"
for t in threads:
    h = C.c_void_p(); msg = C.create_string_buffer(256)
    names_c = (C.c_char_p * 1)(b"chr1"); lens = np.array([248956422], np.int32)
    assert lib.arx_bam_open(out.encode(), 1, names_c, lens.ctypes.data, None, t, -1, C.byref(h), msg, 256) == 0
    reps = 3
    t0 = time.time()
    for _ in range(reps):
        assert lib.arx_bam_write(h, C.byref(b)) == 0
    st = np.zeros(4, np.int64)
    lib.arx_bam_close(h, st.ctypes.data)
    dt = time.time() - t0
    print(f"{t:3d} threads: {reps * n / dt / 1e6:.2f} M records/s, {st[2] / dt / 1e6:.0f} MB/s in, {st[3] / dt / 1e6:.0f} MB/s out, ratio {st[2] / st[3]:.2f}")
os.remove(out)
