#!/bin/bash
# GPU check of a kernel change: the GPU tests, three fuzz modes, the default bench line (kernel times alone)
set -o pipefail
O=gpurun_out/chk; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc: $(tail -n 1 $O/tests.log)"
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/tests.log; then echo "GPU fault in tests"; exit 1; fi
[ $rc -eq 0 ] || { grep -v "^  File" $O/tests.log | tail -30; exit 1; }
for m in mixed repeats long; do
  timeout -k 10 300 python tools/gpu_fuzz.py ${FUZZ_SEEDS:-10} ${FUZZ_BASE:-8200} $m > $O/fuzz_$m.log 2>&1; rc=$?; echo "fuzz $m rc=$rc: $(tail -n 1 $O/fuzz_$m.log)"
  if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/fuzz_$m.log; then echo "GPU fault in fuzz $m"; exit 1; fi
  [ $rc -eq 0 ] || exit 1
done
ARX_AB_ARGS="--steps 10 --warmup 3" bash tools/gpu_bench_ab.sh "ARX_X=1" "ARX_X=2" | cut -c1-330
