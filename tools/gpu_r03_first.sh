#!/bin/bash
# round 3, first GPU call: the new config tests, then bench lines of configs[3] / configs[4] and the default command on this tree
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "config3 or config4" > $O/tests_cfg.log 2>&1; echo "cfg tests rc=$?"; tail -3 $O/tests_cfg.log
timeout -k 10 500 python bench.py --workload vxmix --steps 5 > $O/bench_vxmix.json 2> $O/bench_vxmix.err; echo "vxmix rc=$?"
timeout -k 10 800 python bench.py --workload alt_repeat --steps 5 --cpu-sample 60000 > $O/bench_alt_repeat.json 2> $O/bench_alt_repeat.err; echo "alt_repeat rc=$?"
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
tail -2 $O/*.err
