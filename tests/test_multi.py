"""Several devices behind one handle of the C ABI (arx_multi_*, SURVEY.md s8b's facade): whole barcodes assigned by pair count, one host
thread per device context, the merged result equal -- byte for byte -- to ONE batch over the whole super-batch.  CPU suite: three
contexts of the host test double; -m gpu: two contexts on the one GPU of the box (the code path is the multi-device one; more devices
than the box has cannot be had)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from arachne_amd import api, shard, synth
import workloads

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim", "libarx_hostsim.so")


def _multi_vs_single(lib_path, devices, sizes):
    g = workloads.nasty_genome(47, contig_lens=(150000, 70000), alt_contigs=1)
    d = tempfile.mkdtemp(prefix="arx_multi_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    g.write_alt(fa + ".alt")
    api.index_build(fa, fa, lib_path=lib_path if lib_path != SIM else api.LIB_PATH)
    parts = [synth.make_reads(600 + i, g, 1, n, molecule_len=15000, molecules_per_barcode=3, sub_rate=0.01) for i, n in enumerate(sizes)]
    seqs = np.concatenate([p.seqs for p in parts]); lens = np.concatenate([p.lens for p in parts])
    po = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    flags = np.array([api.worth_running_rfa("A01C01B01D01-1", n) for n in sizes], dtype=np.uint8)
    m = api.MultiReference(fa, devices, lib_path=lib_path)
    try:
        got = m.run(seqs, lens, po, flags)
        again = m.run(seqs, lens, po, flags)                      # the handles are reused (arx_batch_reset)
    finally:
        m.close()
    ref = api.Reference(fa, lib_path=lib_path)
    try:
        b = ref.batch(seqs, lens).run()
        whole = b.fetch()
        c = b.rfa(po, flags)
        b.free()
    finally:
        ref.close()
    for k in ("reg_off", "regs", "alns", "cigars"):
        assert got[k].tobytes() == whole[k].tobytes() == again[k].tobytes(), k
    for k in ("cand_off", "cands"):
        assert got[k].tobytes() == c[k].tobytes() == again[k].tobytes(), k
    expect = shard.lpt_assign(np.diff(po), len(devices))          # the same assignment rule as the RCCL dataflow
    for dev, bcs in enumerate(expect):
        assert (got["device_of_barcode"][bcs] == dev).all()
    return got


def test_three_device_contexts_equal_one_batch_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    got = _multi_vs_single(SIM, [0, 0, 0], [90, 7, 40, 3, 25, 61, 12])
    assert len(set(got["device_of_barcode"].tolist())) == 3 and len(got["regs"]) > 300


def test_more_devices_than_barcodes_hostsim(built):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    _multi_vs_single(SIM, [0, 0, 0, 0], [30, 8])                  # two contexts stay idle


def test_open_failure_is_reported(built):
    with pytest.raises(api.ArachneError, match="arx_multi_open"):
        api.MultiReference("/tmp/no_such_index", [0], lib_path=SIM)


@pytest.mark.gpu
def test_two_contexts_on_the_gpu_equal_one_batch_gpu(built):
    _multi_vs_single(api.LIB_PATH, [0, 0], [300, 40, 120, 9, 77, 201])
