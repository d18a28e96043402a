#!/bin/bash
# bench.py with larger device batches / two steps in flight; run under gpurun
mkdir -p gpurun_out
for cfg in "520000 1" "1001000 1" "350000 2"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 5 --boundary-steps 0 --chunk-pairs $1 --depth $2 > gpurun_out/chunks2_$1_$2.json 2> gpurun_out/chunks2_$1_$2.err || tail -3 gpurun_out/chunks2_$1_$2.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/chunks2_*.json")):
    try:
        d=json.load(open(f)); print(f, d["config"]["device_batches"], "batches", round(d["value"]), "pairs/s", round(d["ms_per_step"],1), "ms/step; seed_bwd iso frac", round(d["roofline"]["isolated"]["frac"],3))
    except Exception as e: print(f, "failed", e)
PY
