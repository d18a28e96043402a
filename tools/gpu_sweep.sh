#!/bin/bash
# launch-shape / batching sweeps of one 350k-pair device batch (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 2>/dev/null | tail -2; }
run X=default
for b in 1 8 32 48; do run ARX_SEED_BATCH=$b; done
for b in 12 16; do run ARX_BPC=$b; done
for b in 16 32 64; do run ARX_COOP_BPC=$b; done
run ARX_EXT_MERGE=0
run ARX_EXT_MERGE=1000000
