#!/bin/bash
# seed kernel times of one 350k-pair batch under launch-parameter variations (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 > /dev/null 2>&1; grep -E "^seed_(fwd|bwd|strat)" gpurun_out/launch_log.tsv | tail -7 | awk '{printf "%s %.2f  ", $1, $3}'; echo; }
run X=default
for b in 16 32 64; do run ARX_SEED_BATCH=$b; done
for c in 16 256; do run ARX_SEED_CHUNK=$c; done
for b in 8 12 20; do run ARX_SEED_BPC=$b ARX_STRAT_BPC=$b; done
for b in 64 96 192 256; do run ARX_SEED_BWD_BUDGET=$b; done
