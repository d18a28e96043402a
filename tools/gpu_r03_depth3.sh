#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not full_step and not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 3 $O/tests.log
[ $rc -eq 0 ] || exit 1
ab() { echo -n "$*: "; env $1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end ${@:2} 2>$O/ab.err > $O/ab.json; rc=$?; if grep -q "HSA_STATUS_ERROR\|out of memory" $O/ab.err; then echo "FAULT/OOM"; tail -n 2 $O/ab.err | cut -c1-300; return 1; fi; python3 -c "
import json,sys; d=json.loads(open('$O/ab.json').read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone sum', round(sum(a.values()),1))"; }
ab X=1 --depth 2 --warmup 2 --steps 12 || exit 1
ab X=1 --depth 3 --warmup 3 --steps 12 || exit 1
ab X=1 --depth 4 --warmup 4 --steps 12 || exit 1
ab ARX_KMER_K=15 --depth 4 --warmup 4 --steps 12
