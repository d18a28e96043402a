#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not full_step and not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 3 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/gpu_fuzz.py 40 10000 mixed > $O/fuzz.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -n 1 $O/fuzz.log
[ $rc -eq 0 ] || exit 1
ab() { echo -n "$*: "; env $1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end ${@:2} 2>$O/ab.err > $O/ab.json; rc=$?; if grep -q "HSA_STATUS_ERROR\|out of memory" $O/ab.err; then echo "FAULT/OOM"; tail -n 2 $O/ab.err | cut -c1-300; return 1; fi; python3 -c "
import json,sys; d=json.loads(open('$O/ab.json').read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone sum', round(sum(a.values()),1), 'fwd', a.get('seed_fwd'), 'bwd', a.get('seed_bwd'), 'fwd frac', round(d['roofline_fwd']['frac'],3), 'stage', round(d['roofline_seeding_stage']['frac'],3))"; }
ab ARX_KMER_FWD=0 || exit 1
ab ARX_KMER_FWD=1 || exit 1
