"""bench.py's N > 1 path on the CPU: two ranks under torch.distributed.run with the gloo backend, the host test double in
place of the GPU library.  Barcode groups are independent, so ranks share nothing on the data path: each aligns its own
read set; torch.distributed carries only the barriers, the max-over-ranks time and the per-rank bookkeeping."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "hostsim", "libarx_hostsim.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(tmp, n):
    args = ["bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0", "--lib", SIM, "--backend", "gloo", "--genome-len", "300000",
            "--barcodes", "3", "--pairs-per-barcode", "40", "--chunk-pairs", "80", "--cache", str(tmp), "--no-cpu-baseline", "--scatter-steps", "1"]
    if n == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # ONE json line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_shard_by_barcode_groups(built, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(SIM)])
    one = _bench(tmp_path, 1)
    two = _bench(tmp_path, 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert [r["rank"] for r in two["per_rank"]] == [0, 1]
    assert two["per_rank"][0]["read_seed"] != two["per_rank"][1]["read_seed"]          # every rank has its own barcodes
    assert two["per_rank"][0] == one["per_rank"][0]                                    # rank 0 does exactly the N = 1 work
    assert two["config"]["pairs_per_step_per_gpu"] == one["config"]["pairs_per_step_per_gpu"] == 120
    assert abs(two["value"] * two["ms_per_step"] / 1000.0 - 240) < 1e-6                # value = pairs of ALL ranks / max-over-ranks time
    for k in ("metric", "unit", "steps", "warmup", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in two
    # the scatter / gather dataflow ran as well: rank 0 assigned whole barcodes of both ranks' read sets (6 barcodes x 40 pairs) by pair count
    assert sorted(two["scatter_gather"]["pairs_per_rank"]) == [120, 120] and "scatter_gather" not in one
    assert two["boundary"]["value"] > 0
