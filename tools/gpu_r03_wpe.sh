#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
ab() { echo -n "$1: "; env $1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end 2>$O/ab.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['kernel_ms_per_step_alone']; print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; alone: chain', a.get('chain'), 'ext_step', a.get('ext_step'), 'mapq', a.get('mapq'), 'ext_init', a.get('ext_init'), 'sum', round(sum(a.values()),1))"; }
ab ARX_LIB=$PWD/arachne_amd/libarachne_amd.so
ab ARX_LIB=$PWD/arachne_amd/variants/lib_wpe3.so
ab ARX_LIB=$PWD/arachne_amd/variants/lib_wpe4.so
ARX_SA_DENSE=4 timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end --steps 3 > $O/sa4.json 2> $O/sa4.err; echo "sa4 rc=$?"; tail -n 3 $O/sa4.err
python -m pytest tests/test_multi.py tests/test_e2e.py -m gpu -q 2>&1 | tail -n 2
