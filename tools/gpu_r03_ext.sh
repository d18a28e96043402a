#!/bin/bash
# the round-3 extension kernel: GPU suite, fuzz, A/B against round 2's kernel on the default command
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not full_step and not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python tools/gpu_fuzz.py 40 3000 mixed > $O/fuzz_mixed.log 2>&1; rc=$?; echo "fuzz mixed rc=$rc"; tail -n 2 $O/fuzz_mixed.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python tools/gpu_fuzz.py 30 4000 repeats > $O/fuzz_rep.log 2>&1; rc=$?; echo "fuzz repeats rc=$rc"; tail -n 2 $O/fuzz_rep.log
[ $rc -eq 0 ] || exit 1
ab() { echo -n "$1 : "; env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone: extend', a.get('extend'), 'ext_step', a.get('ext_step'), 'ext_init', a.get('ext_init'), 'sum', round(sum(a.values()),1))"; }
ab ARX_EXT_OLD=1
ab ARX_EXT_OLD=0
ab ARX_EXT_CHUNK=4
ab ARX_EXT_CHUNK=64
