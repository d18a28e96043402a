#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
ab() { echo -n "$*: "; env $1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end ${@:2} 2>$O/ab.err > $O/ab.json; rc=$?; if grep -q HSA_STATUS_ERROR $O/ab.err; then echo "GPU FAULT"; exit 1; fi; python3 -c "
import json,sys; d=json.loads(open('$O/ab.json').read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone sum', round(sum(a.values()),1), 'roofline', round(d['roofline']['frac'],3), round(d['roofline'].get('isolated',{}).get('frac',0),3))"; }
ab X=1 --chunk-pairs 1001000 --depth 2 || exit 1
ab X=1 --chunk-pairs 1001000 --depth 2 --steps 20
ab X=1 --chunk-pairs 1001000 --depth 3 --steps 12
ab X=1 --chunk-pairs 1001000 --depth 4 --steps 12
ab X=1 --workload alt_repeat --steps 6 --chunk-pairs 1001000 --depth 2
ab X=1 --workload vxmix --chunk-pairs 1001000 --depth 2
ab X=1 --workload chr20 --chunk-pairs 1001000 --depth 2
