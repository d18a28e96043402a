#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not full_step and not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/gpu_fuzz.py 30 7000 repeats > $O/fuzz_rep.log 2>&1; rc=$?; echo "fuzz repeats rc=$rc"; tail -n 1 $O/fuzz_rep.log
[ $rc -eq 0 ] || exit 1
ab() { echo -n "$1 $2: "; env $1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end --workload $2 --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['kernel_ms_per_step_alone']; t=d['kernel_ms_per_step']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone/timed: rescue_heavy', a.get('rescue_heavy'), t.get('rescue_heavy'), 'chain_heavy', a.get('chain_heavy'), t.get('chain_heavy'), 'sum alone', round(sum(a.values()),1))"; }
ab ARX_RESCUE_LDS_CLASSES=0 alt_repeat
ab ARX_RESCUE_LDS_CLASSES=1 alt_repeat
ab ARX_RESCUE_LDS_CLASSES=1 grch38
ab ARX_RESCUE_LDS_CLASSES=1 vxmix
