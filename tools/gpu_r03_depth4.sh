#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
ab() { echo -n "$*: "; env $1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end ${@:2} 2>$O/ab.err > $O/ab.json; rc=$?; if grep -q "HSA_STATUS_ERROR\|out of memory" $O/ab.err; then echo "FAULT/OOM"; tail -n 2 $O/ab.err | cut -c1-300; return 1; fi; python3 -c "
import json,sys; d=json.loads(open('$O/ab.json').read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone sum', round(sum(a.values()),1), d['setup_s'])"; }
ab X=1 || exit 1
ab X=1 --no-stagger --streams 3 --steps 12
ab X=1 --steps 30
ab X=1 --workload chr20
ab X=1 --workload vxmix
