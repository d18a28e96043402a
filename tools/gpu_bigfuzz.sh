# seeds start at 20000 / 21000 / 22000 + ARX_FUZZ_BASE (a fresh base per campaign)
B=${ARX_FUZZ_BASE:-0}
mkdir -p gpurun_out/bigfuzz
for spec in "mixed 80 20000" "repeats 60 21000" "long 60 22000"; do
  set -- $spec
  timeout -k 10 380 python tools/gpu_fuzz.py $2 $(($3 + B)) $1 > gpurun_out/bigfuzz/$1.log 2>&1; rc=$?
  echo "fuzz $1 ($2 seeds from $(($3 + B))) rc=$rc: $(tail -n 1 gpurun_out/bigfuzz/$1.log); ok lines: $(grep -c ' ok:' gpurun_out/bigfuzz/$1.log)"
  if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/bigfuzz/$1.log; then echo "GPU fault"; exit 1; fi
  [ $rc -eq 0 ] || { grep MISMATCH gpurun_out/bigfuzz/$1.log | head -5; exit 1; }
done
