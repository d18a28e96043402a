#!/bin/bash
# GPU check of text mode (whole suffix array + inverse): GPU tests, a mixed fuzz, then the default bench line with and without it on the same box.
set -o pipefail
mkdir -p gpurun_out/r03t
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03t/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03t/tests.log
if grep -q HSA_STATUS_ERROR gpurun_out/r03t/tests.log; then echo "GPU fault in tests"; exit 1; fi
[ $rc -eq 0 ] || { tail -40 gpurun_out/r03t/tests.log; exit 1; }
timeout -k 10 300 python tools/gpu_fuzz.py 12 4000 mixed > gpurun_out/r03t/fuzz_mixed.log 2>&1; rc=$?; echo "fuzz mixed rc=$rc"; tail -2 gpurun_out/r03t/fuzz_mixed.log
[ $rc -eq 0 ] || exit 1
bash tools/gpu_bench_ab.sh "ARX_TEXT_INDEX=1" "ARX_TEXT_INDEX=0" > gpurun_out/r03t/ab.log 2>&1; rc=$?; cat gpurun_out/r03t/ab.log
[ $rc -eq 0 ] || exit 1
