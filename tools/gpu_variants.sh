#!/bin/bash
# run the dedup probe with every library variant, then the full stage probe with those that pass
mkdir -p gpurun_out
for v in arachne_amd/variants/*.so; do
  ARX_LIB=$PWD/$v timeout -k 5 25 python tools/gpu_probe2.py 0 64 > gpurun_out/var.out 2>&1; rc=$?
  echo "$v dedup rc=$rc $(tail -1 gpurun_out/var.out)"
  if [ $rc -eq 0 ]; then
    echo "== full probe with $v"
    ARX_LIB=$PWD/$v timeout -k 5 60 python tools/gpu_probe.py 600 2>&1 | tail -14
  fi
done
