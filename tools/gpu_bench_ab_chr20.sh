#!/bin/bash
# A/B of run-time switches on the chr20 workload (run under gpurun)
run() { echo -n "$* : "; env "$@" timeout -k 10 300 python3 bench.py --workload chr20 --no-cpu-baseline --boundary-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step')"; }
for s in "$@"; do run $s; done
