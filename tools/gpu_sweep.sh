#!/bin/bash
# launch-shape / batching sweeps of one 350k-pair device batch (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 2>/dev/null | tail -2; }
run X=default
for b in 8 16; do for c in 32 256; do run ARX_BPC=$b ARX_SEED_CHUNK=$c; done; done
run ARX_BPC=16 ARX_SEED_CHUNK=64
run ARX_SEED_CHUNK=16
for b in 32 128 256; do run ARX_COOP_BPC=$b; done
run ARX_EXT_MERGE=10000
run ARX_EXT_MERGE=100000
