"""CPU differential test: oracle restatement vs the reference's own C core compiled in place
(oracle/_ref/libbwaref.so) on freshly seeded nasty workloads.  Skipped where _ref is absent."""
import os
import tempfile

import numpy as np
import pytest

import refdrv
import workloads

pytestmark = pytest.mark.skipif(not refdrv.available(), reason="oracle/_ref/libbwaref.so not built")


@pytest.mark.parametrize("seed", [1, 2])
def test_pair_path_matches_reference(built, seed):
    import oradrv
    g = workloads.nasty_genome(seed, contig_lens=(120000, 70000, 30000), alt_contigs=2)
    rs = workloads.nasty_reads(seed, g, n_barcodes=4, pairs_per_barcode=400)
    tmp = tempfile.mkdtemp(prefix="arx_diff_")
    prefix = os.path.join(tmp, "g.fa")
    g.write_fasta(prefix)
    g.write_alt(prefix + ".alt")
    r = refdrv.Ref()
    r.index_build(prefix, prefix)
    r.open(prefix)
    o = oradrv.Oracle(prefix)
    A = r.batch(rs.seqs, rs.lens, n_threads=2)
    B = o.batch(rs.seqs, rs.lens, n_threads=2)
    for key in ("reg_off", "regs", "alns", "cigars"):
        assert A[key].shape == B[key].shape and (A[key] == B[key]).all(), key
    assert A["regs"].shape[0] > 2000
    r.close()
    o.close()
