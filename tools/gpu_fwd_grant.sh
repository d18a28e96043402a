#!/bin/bash
# forward kernel launch times by grant threshold (run under gpurun): one 333 k-pair GRCh38-size batch alone
for g in 64 16 4 1; do
  echo -n "ARX_SEED_GRANT=$g : "
  ARX_SEED_GRANT=$g ROUNDS_BRIEF=1 timeout -k 10 200 python3 tools/gpu_rounds.py 4333 grch38 > /dev/null 2>&1
  grep -E "^seed_fwd" gpurun_out/launch_log.tsv | tail -2 | awk '{printf "%s items %.3f ms   ", $2, $3}'; echo
done
