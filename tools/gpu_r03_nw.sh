#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not full_step and not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/gpu_fuzz.py 30 9000 mixed > $O/fuzz.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -n 1 $O/fuzz.log
[ $rc -eq 0 ] || exit 1
ab() { echo -n "$1 $2: "; env $1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end --workload $2 --steps ${3:-10} 2>$O/ab.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone: nw', a.get('reg2aln_nw'), 'locate', a.get('locate'), 'sum', round(sum(a.values()),1), 'open_s', d['setup_s'].get('arx_open_s'))"; }
ab ARX_SA_DENSE=8 grch38
ab ARX_SA_DENSE=4 grch38
ab ARX_SA_DENSE=8 alt_repeat 5
