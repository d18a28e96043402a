#!/bin/bash
# backward-sweep reservation size (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 > /dev/null 2>&1; grep -E "^seed_(fwd|bwd|strat)" gpurun_out/launch_log.tsv | tail -7 | awk '{printf "%s %.2f  ", $1, $3}'; echo; }
for c in 64 32 16 8 4; do run ARX_SEED_BWD_CHUNK=$c; done
run ARX_SEED_BWD_CHUNK=16 ARX_SEED_BATCH=64
run ARX_SEED_BWD_CHUNK=16 ARX_SEED_BPC=12
