#!/bin/bash
# lane-parking threshold of the seeding kernels at GRCh38 size (run under gpurun): seed stats + per-batch kernel times
for b in 48 32 16 8; do
  echo "== ARX_SEED_BATCH=$b"
  ARX_SEED_BATCH=$b ARX_SEED_STATS=1 ROUNDS_BRIEF=1 python3 tools/gpu_rounds.py 4333 grch38 2>&1 | grep -E "seed stats|^\{" | tail -5 | cut -c1-420
done
