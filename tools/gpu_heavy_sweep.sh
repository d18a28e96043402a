#!/bin/bash
# thresholds of the wavefront-per-item kernels (run under gpurun): one 333 k-pair GRCh38-size batch alone, per-kernel ms
run() { echo "== $*"; env "$@" ROUNDS_BRIEF=1 timeout -k 10 200 python3 tools/gpu_rounds.py 4333 grch38 2>/dev/null | grep -E "wall|^\{" | python3 -c "
import sys, ast
for l in sys.stdin:
    if l.startswith('wall'): print(l.strip())
    else:
        d = ast.literal_eval(l.strip()); print({k: d[k] for k in ('chain', 'chain_heavy', 'dedup', 'dedup_heavy', 'rescue_step', 'rescue_heavy') if k in d})
"; }
run X=0
run ARX_CHAIN_HEAVY_MIN=32
run ARX_CHAIN_HEAVY_MIN=20
run ARX_DEDUP_HEAVY_MIN=16
run ARX_DEDUP_HEAVY_MIN=8
run ARX_RESCUE_HEAVY_MIN=24
run ARX_RESCUE_HEAVY_MIN=12
