"""CPU test: the product's index builder (arx_index_build, SA-IS) writes files byte-identical to the reference's
`bwa index` -- against the golden index (built by the reference's code) and, where oracle/_ref is present, on fresh genomes."""
import filecmp
import os
import tempfile

import numpy as np
import pytest

import __graft_entry__ as ge
import refdrv
import workloads
from arachne_amd import api, synth

GOLD = os.path.join(workloads.GOLDEN_DIR, "bwa_path_v1.npz")


def _genome_from_gold(z):
    lens = z["genome_lens"]
    cat = z["genome_cat"]
    off = np.concatenate([[0], np.cumsum(lens)])
    seqs = [cat[off[i]:off[i + 1]] for i in range(len(lens))]
    names = [f"chrS{i + 1}" for i in range(len(lens) - 1)] + ["chrS1_alt1"]
    return synth.Genome(names, seqs, [False] * len(seqs))


def test_matches_golden_index_bytes():
    ge.build_product()
    z = np.load(GOLD)
    d = tempfile.mkdtemp(prefix="arx_idx_")
    gold_prefix = workloads.unpack_index(z, d, "gold.fa")
    fa = os.path.join(d, "our.fa")
    _genome_from_gold(z).write_fasta(fa)
    api.index_build(fa, fa)          # host-side entry of libarachne_amd.so; needs no GPU
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(gold_prefix + "." + ext, fa + "." + ext, shallow=False), ext


@pytest.mark.skipif(not refdrv.available(), reason="oracle/_ref/libbwaref.so not built")
@pytest.mark.parametrize("lens", [(1000003, 777, 50021), (64, 19), (300000,)])
def test_matches_reference_builder(lens):
    ge.build_product()
    g = workloads.nasty_genome(5, contig_lens=lens, alt_contigs=0) if min(lens) > 1000 else synth.make_genome(5, list(lens), repeat_families=[], n_runs=0)
    d = tempfile.mkdtemp(prefix="arx_idx_")
    fa = os.path.join(d, "g.fa")
    g.write_fasta(fa)
    refdrv.Ref().index_build(fa, os.path.join(d, "ref"))
    api.index_build(fa, os.path.join(d, "our"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(os.path.join(d, "ref." + ext), os.path.join(d, "our." + ext), shallow=False), ext


def test_errors_are_reported():
    ge.build_product()
    with pytest.raises(api.ArachneError):
        api.index_build("/nonexistent.fa", "/tmp/x")
