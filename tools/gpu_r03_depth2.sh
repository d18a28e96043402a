#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
ab() { echo -n "$*: "; env $1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end ${@:2} 2>$O/ab.err > $O/ab.json; rc=$?; if grep -q HSA_STATUS_ERROR $O/ab.err; then echo "GPU FAULT"; tail -n 2 $O/ab.err; exit 1; fi; python3 -c "
import json,sys; d=json.loads(open('$O/ab.json').read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone sum', round(sum(a.values()),1), 'strat', a.get('seed_strat'))"; }
ab ARX_KMER_K=15 --depth 3 --warmup 3 --steps 12 || exit 1
ab ARX_KMER_K=14 --depth 4 --warmup 4 --steps 12 || exit 1
ab ARX_KMER_K=14 --depth 3 --warmup 3 --steps 15 --chunk-pairs 700000
ab ARX_KMER_K=14 --depth 2 --warmup 4 --steps 12 --chunk-pairs 520000
