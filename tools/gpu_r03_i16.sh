#!/bin/bash
# GPU check of the 250-255-base read path: the GPU tests, the long-read fuzz, a mixed fuzz, and the default bench line (no regression).
set -o pipefail
mkdir -p gpurun_out/r03i
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03i/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03i/tests.log
if grep -q HSA_STATUS_ERROR gpurun_out/r03i/tests.log; then echo "GPU fault in tests"; exit 1; fi
[ $rc -eq 0 ] || { tail -40 gpurun_out/r03i/tests.log; exit 1; }
timeout -k 10 400 python tools/gpu_fuzz.py 24 900 long > gpurun_out/r03i/fuzz_long.log 2>&1; rc=$?; echo "fuzz long rc=$rc"; tail -3 gpurun_out/r03i/fuzz_long.log
if grep -q HSA_STATUS_ERROR gpurun_out/r03i/fuzz_long.log; then echo "GPU fault in fuzz"; exit 1; fi
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/gpu_fuzz.py 12 3000 mixed > gpurun_out/r03i/fuzz_mixed.log 2>&1; rc=$?; echo "fuzz mixed rc=$rc"; tail -2 gpurun_out/r03i/fuzz_mixed.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-end-to-end > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err; rc=$?; echo "bench rc=$rc"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03i/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["parity_ok"], d["kernel_ms_per_step_alone"])
PY
