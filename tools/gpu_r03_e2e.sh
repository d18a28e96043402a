#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not full_step and not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/gpu_fuzz.py 25 6000 mixed > $O/fuzz_mixed.log 2>&1; rc=$?; echo "fuzz mixed rc=$rc"; tail -n 1 $O/fuzz_mixed.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); a=d['kernel_ms_per_step_alone']
print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone: extend', a.get('extend'), 'sw_u8', a.get('sw_u8'), 'nw', a.get('reg2aln_nw'), 'sum', round(sum(a.values()),1)); print('e2e', d.get('end_to_end')); print('parity', d.get('parity_ok'), 'boundary', d['boundary']['value'], d['boundary'].get('matches_resident'))"
timeout -k 10 300 python tools/bam_sink_bench.py 400000 1 4 8 16 32 64 > $O/bam_sink_bench.txt 2>&1; cat $O/bam_sink_bench.txt
