#!/bin/bash
# launch parameters of the backward-sweep kernel alone (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 > /dev/null 2>&1; grep -E "^seed_(fwd|bwd|strat)" gpurun_out/launch_log.tsv | tail -7 | awk '{printf "%s %.2f  ", $1, $3}'; echo; }
run X=default
for b in 32 40 56 64; do run ARX_SEED_BWD_BATCH=$b; done
for b in 160 192 256; do run ARX_SEED_BWD_BUDGET=$b; done
for c in 24 48; do run ARX_SEED_BWD_CHUNK=$c; done
