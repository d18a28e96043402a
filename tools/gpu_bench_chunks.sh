#!/bin/bash
# bench.py at several device-batch sizes (same step, more batches in flight); run under gpurun
mkdir -p gpurun_out
python3 bench.py --no-cpu-baseline --steps 5 > gpurun_out/chunks_350.json 2> gpurun_out/chunks_350.err
for c in 250000 170000 120000 85000; do
  python3 bench.py --no-cpu-baseline --steps 5 --chunk-pairs $c > gpurun_out/chunks_$c.json 2> gpurun_out/chunks_$c.err || tail -3 gpurun_out/chunks_$c.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/chunks_*.json")):
    try:
        d=json.load(open(f)); print(f, d["config"]["device_batches"], "batches", round(d["value"]), "pairs/s", round(d["ms_per_step"],1), "ms/step; seed_bwd iso frac", round(d["roofline"]["isolated"]["frac"],3))
    except Exception as e: print(f, "failed", e)
PY
