#!/bin/bash
# seeding kernel times of one 350k-pair batch for builds with different register budgets (tools/build_variants.sh wN "-DARX_SEED_WPE=N")
# and resident-block counts (diagnostics; run under gpurun)
export ROUNDS_BRIEF=1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_rounds.py 350 > gpurun_out/wpe_last.txt 2>&1; grep -E "^seed_(fwd|bwd|strat)" gpurun_out/launch_log.tsv | tail -7 | awk '{printf "%s %.2f  ", $1, $3}'; echo; }
V=$PWD/arachne_amd/variants
run ARX_LIB=$V/lib_w4.so ARX_SEED_BPC=16 ARX_STRAT_BPC=16
run ARX_LIB=$V/lib_w4.so ARX_SEED_BPC=20 ARX_STRAT_BPC=24
run ARX_LIB=$V/lib_w4.so ARX_SEED_BPC=24 ARX_STRAT_BPC=28
run ARX_LIB=$V/lib_w5.so
run ARX_LIB=$V/lib_w6.so
run ARX_LIB=$V/lib_w6.so ARX_SEED_BATCH=40
run ARX_LIB=$V/lib_w8.so
run ARX_LIB=$V/lib_w8.so ARX_SEED_BATCH=40
