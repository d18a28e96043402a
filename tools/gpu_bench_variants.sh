#!/bin/bash
# bench.py under a few batch/stream shapes (diagnostics; run under gpurun)
b() { echo "== $*"; timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step', 'frac', round(d.get('roofline',{}).get('frac',0),4))"; }
b
b --chunk-pairs 250000 --streams 4
b --chunk-pairs 200000 --streams 5
b --chunk-pairs 170000 --streams 6
b --chunk-pairs 500000 --streams 2
b --chunk-pairs 350000 --streams 2
b --chunk-pairs 1000000 --streams 1
