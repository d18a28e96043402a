"""Worker of tests/test_shard_scatter.py (launched by torch.distributed.run, gloo, the host test double in place of the GPU library):
rank 0 owns the read set, assigns whole barcodes by pair count (LPT), scatters the packed batches, every rank runs its batch, rank 0
gathers the slabs, renumbers them into read-set order and writes them next to the N = 1 result of one batch over everything."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from arachne_amd import api, shard, synth
    import workloads
    out_dir, lib = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "host"                     # "device": payloads as tensors the library reads / writes in place
    sizes_arg = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else None
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prefix = os.path.join(out_dir, "g.fa")
    assign = pair_off = None
    if rank == 0:
        g = workloads.nasty_genome(41, contig_lens=(120000, 60000), alt_contigs=1)
        g.write_fasta(prefix)
        g.write_alt(prefix + ".alt")
        api.index_build(prefix, prefix)
        # barcodes of very different sizes, so that the LPT assignment is not the trivial split
        sizes = sizes_arg or [90, 7, 40, 3, 25, 61, 12]
        parts = [synth.make_reads(500 + i, g, 1, n, molecule_len=15000, molecules_per_barcode=3 if n < 1000 else 40, sub_rate=0.01) for i, n in enumerate(sizes)]
        seqs = np.concatenate([p.seqs for p in parts]); lens = np.concatenate([p.lens for p in parts])
        pair_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        do_rfa = np.array([api.worth_running_rfa("A01C01B01D01-1", n) for n in sizes], dtype=np.uint8)
        assign = shard.lpt_assign(np.diff(pair_off), world)
        packed = [shard.pack(seqs, lens, pair_off, do_rfa, a) for a in assign]
    else:
        packed = None
    dist.barrier()                                       # the index files exist
    ref = api.Reference(prefix, lib_path=lib)
    xch = shard.Exchange(dist, "cpu")
    if mode == "device":
        gathered, _h = shard.step_device(xch, rank, world, ref, packed)
        if world > 1:                                                       # a second step through the same handles (reset_device on a used batch)
            gathered, _h = shard.step_device(xch, rank, world, ref, packed, _h)
    else:
        mine = shard.scatter_batches(xch, rank, world, packed)
        res, _h = shard.run_batch(ref, mine)
        gathered = shard.gather_results(xch, rank, world, res)
    if rank == 0:
        merged = shard.merge_in_read_set_order(gathered, assign, pair_off)
        whole, _h2 = shard.run_batch(ref, shard.pack(seqs, lens, pair_off, do_rfa, np.arange(len(sizes))))   # N = 1: one batch over everything
        np.savez(os.path.join(out_dir, "result.npz"), loads=np.array([int(np.diff(pair_off)[a].sum()) for a in assign]),
                 **{"m_" + k: v for k, v in merged.items()}, **{"w_" + k: v for k, v in whole.items()})
    ref.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
