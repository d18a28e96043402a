#!/bin/bash
# rocprofv3 passes over the default bench command (run under gpurun).  Kernel trace, then separate --pmc passes
# (never combined with a trace domain), then the FETCH_SIZE calibration microbench.  Summaries: tools/prof_summary.py.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --boundary-steps 0 ${ARX_PROFILE_ARGS:-}"   # the default command (GRCh38-size workload, K = 10 timed steps after 2 warm-up steps) minus the CPU leg and the boundary pass
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err   # also warms the index cache in /tmp
echo "plain rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc1.json 2> $OUT/pmc1.err
echo "pmc fetch rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_pmc2.json 2> $OUT/pmc2.err
echo "pmc sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/bench_pmc3.json 2> $OUT/pmc3.err
echo "pmc sq2 rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o $OUT/calib_fetch && \
  $OUT/calib_fetch > $OUT/calib_plain.txt 2>&1 && \
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -- $OUT/calib_fetch > $OUT/calib_pmc.txt 2>&1
echo "calib rc=$?"
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +20M -delete   # keep the merge-back under the size limit
rm -f $OUT/calib_fetch
du -sh $OUT
