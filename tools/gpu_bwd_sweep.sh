#!/bin/bash
# resident blocks per CU of the seeding kernels (run under gpurun): backward sweeps alone (roofline.isolated) and throughput, default workload
run() { echo -n "$* : "; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --steps 4 ${WL:-} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), 'pairs/s; bwd alone', round(r['isolated']['frac'],3), round(r['isolated']['avg_launch_ms'],2), 'ms; timed', round(r['frac'],3), '; fwd', round(d['roofline_fwd']['frac'],3))"; }
for s in "$@"; do run $s; done
