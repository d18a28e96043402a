#!/bin/bash
# GPU check of the 21-lane bin of the backward sweeps: GPU tests, a fuzz, then A/B of the bin's upper bound on the default bench line
set -o pipefail
mkdir -p gpurun_out/r03m
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03m/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03m/tests.log
if grep -q HSA_STATUS_ERROR gpurun_out/r03m/tests.log; then echo "GPU fault in tests"; exit 1; fi
[ $rc -eq 0 ] || { tail -40 gpurun_out/r03m/tests.log; exit 1; }
timeout -k 10 300 python tools/gpu_fuzz.py 10 5000 mixed > gpurun_out/r03m/fuzz_mixed.log 2>&1; rc=$?; echo "fuzz mixed rc=$rc"; tail -2 gpurun_out/r03m/fuzz_mixed.log
[ $rc -eq 0 ] || exit 1
ARX_AB_ARGS="--steps 8 --warmup 2" bash tools/gpu_bench_ab.sh "ARX_SEED_BWD_MID=21" "ARX_SEED_BWD_MID=16" "ARX_SEED_BWD_MID=20" "ARX_SEED_BWD_MID=19" > gpurun_out/r03m/ab.log 2>&1; rc=$?; cut -c1-330 gpurun_out/r03m/ab.log
ARX_SEED_STATS=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end --steps 1 --warmup 1 2>&1 >/dev/null | grep "backward tasks" | sort | uniq -c | head -4
[ $rc -eq 0 ] || exit 1
