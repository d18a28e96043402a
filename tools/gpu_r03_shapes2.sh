#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
ab() { echo -n "$*: "; env $1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end ${@:2} 2>$O/ab.err > $O/ab.json; rc=$?; if grep -q HSA_STATUS_ERROR $O/ab.err; then echo "GPU FAULT"; exit 1; fi; python3 -c "
import json,sys; d=json.loads(open('$O/ab.json').read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone sum', round(sum(a.values()),1), 'locate frac', round(d['roofline_locate']['frac'],3))"; }
ab X=1 || exit 1
ab X=1 --chunk-pairs 1001000 --streams 3 --depth 3
ab X=1 --chunk-pairs 520000 --streams 3 --depth 2
ab X=1 --chunk-pairs 1001000 --streams 2 --depth 2
