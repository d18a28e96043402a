"""The multi-GPU dataflow of SURVEY.md s8e on the CPU: world_size 2 over gloo with the host test double -- LPT assignment of whole
barcodes, scatter of packed batches, gather of result slabs; the gathered output equals the N = 1 output byte for byte."""
import os
import socket
import subprocess
import sys

import numpy as np

from arachne_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "hostsim", "libarx_hostsim.so")


def test_lpt_assignment_is_greedy_by_pair_count_and_deterministic():
    a = shard.lpt_assign([90, 7, 40, 3, 25, 61, 12], 2)
    assert [x.tolist() for x in a] == [[0, 3, 4], [1, 2, 5, 6]]           # 90 | 61, 40 | 25 -> rank 0 (115 vs 101) ... loads 118 / 120
    loads = [sum([90, 7, 40, 3, 25, 61, 12][i] for i in x) for x in a]
    assert loads == [118, 120]
    assert [x.tolist() for x in shard.lpt_assign([5, 5, 5], 4)] == [[0], [1], [2], []]
    one = shard.lpt_assign([30000, 201, 77], 1)
    assert one[0].tolist() == [0, 1, 2]


def _run_workers(tmp_path, world, mode="host", sizes=None, timeout=900):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "shard_worker.py"), str(tmp_path), SIM, mode] + ([",".join(str(x) for x in sizes)] if sizes else [])
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    z = np.load(os.path.join(str(tmp_path), "result.npz"))
    for k in ("reg_off", "regs", "alns", "cigars", "cand_off", "cands"):
        m, w = z["m_" + k], z["w_" + k]
        assert m.dtype == w.dtype and m.shape == w.shape and m.tobytes() == w.tobytes(), k
    return z


def test_scatter_gather_over_gloo_equals_single_batch(built, tmp_path):
    z = _run_workers(tmp_path, 2)
    assert z["loads"].tolist() == [118, 120]
    assert len(z["w_regs"]) > 400 and len(z["w_cands"]) >= len(z["w_regs"])


def test_device_resident_scatter_gather_equals_single_batch(built, tmp_path):
    """The same dataflow with payloads the library reads and writes in place (arx_batch_reset_device / arx_batch_device_view, transfers posted
    together with batch_isend_irecv), two steps through the same batch handles."""
    z = _run_workers(tmp_path, 2, mode="device")
    assert z["loads"].tolist() == [118, 120]


def test_world_4_with_an_empty_rank_and_a_30000_pair_barcode(built, tmp_path):
    """Four ranks, three barcodes: the largest set the reader returns (30,000 pairs, reader.go:236) takes a rank by itself, one rank gets
    nothing at all; gathered == one batch over everything, byte for byte (device-resident path)."""
    z = _run_workers(tmp_path, 4, mode="device", sizes=[30000, 201, 77], timeout=1500)
    assert sorted(z["loads"].tolist()) == [0, 77, 201, 30000]
    assert len(z["w_reg_off"]) == 2 * 30278 + 1
