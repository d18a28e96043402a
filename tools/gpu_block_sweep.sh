#!/bin/bash
# RFA kernel time of one 350k-pair batch for builds with 256 / 512 / 1024 lanes per barcode workgroup (tools/build_variants.sh bN "-DARX_BLOCK_LANES=N")
export ROUNDS_BRIEF=1
V=$PWD/arachne_amd/variants
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 > gpurun_out/blk_last.txt 2>&1; grep -E "^(rfa|mapq|cand_build)" gpurun_out/launch_log.tsv | awk '{printf "%s %.2f  ", $1, $3}'; echo; tail -1 gpurun_out/blk_last.txt; }
run X=default
run ARX_LIB=$V/lib_b512.so
run ARX_LIB=$V/lib_b1024.so
