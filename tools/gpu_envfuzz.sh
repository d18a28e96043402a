mkdir -p gpurun_out/envfuzz
i=0
for cfg in "ARX_TEXT_INDEX=0" "ARX_SEED_BWD2=3" "ARX_CHAIN_MID_MIN=0" "ARX_KMER_FWD=0" "ARX_TEXT_BWD=0 ARX_SEED_FIT32=0" "ARX_SEED_BWD_MID=16" "ARX_SW_SIMPLE=1"; do
  i=$((i+1))
  env $cfg timeout -k 10 200 python tools/gpu_fuzz.py 8 $((30000 + 100 * i)) mixed > gpurun_out/envfuzz/$i.log 2>&1; rc=$?
  echo "$cfg: rc=$rc $(tail -n 1 gpurun_out/envfuzz/$i.log)"
  if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/envfuzz/$i.log; then echo "GPU fault"; exit 1; fi
  [ $rc -eq 0 ] || { grep MISMATCH gpurun_out/envfuzz/$i.log | head -3; exit 1; }
done
