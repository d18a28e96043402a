#!/bin/bash
# reads per seeding group at GRCh38 size (run under gpurun)
for g in 0 360000 180000 90000; do
  echo "== ARX_SEED_GROUP=$g"
  ARX_SEED_GROUP=$g ROUNDS_BRIEF=1 python3 tools/gpu_rounds.py 4333 grch38 2>&1 | grep -E "^\{|wall" | tail -2 | cut -c1-400
done
