#!/bin/bash
# rocprofv3 passes over a reduced bench (100 barcodes = 100k pairs, one stream so kernels do not overlap)
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
ARGS="bench.py --streams 1 --barcodes 100 --steps 1 --warmup 1 --no-cpu-baseline"
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err   # warms the index cache in /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc1.json 2> $OUT/pmc1.err
echo "pmc fetch rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_pmc2.json 2> $OUT/pmc2.err
echo "pmc sq rc=$?"
find $OUT -name "*.csv" | head -20
du -sh $OUT
