#!/bin/bash
# seed kernel times of one 350k-pair batch under launch-parameter variations (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 > /dev/null 2>&1; grep -E "^seed_(fwd|bwd|strat)" gpurun_out/launch_log.tsv | tail -5 | tr '\n' ' '; echo; }
run X=default
for b in 1 32 48 64; do run ARX_SEED_BATCH=$b; done
for c in 8 16 256; do run ARX_SEED_CHUNK=$c; done
for b in 4 8 12 19; do run ARX_BPC=$b; done
