#!/bin/bash
# compile-time waves-per-SIMD budget of the seeding kernels (variants under arachne_amd/variants/, run under gpurun)
run() { lib=$1; shift; echo -n "$lib $* : "; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --steps 4 --lib $lib 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), 'pairs/s; bwd alone', round(r['isolated']['frac'],3), round(r['isolated']['avg_launch_ms'],2), 'ms; timed', round(r['frac'],3), '; fwd', round(d['roofline_fwd']['frac'],3), 'strat', round(d['roofline_strat']['frac'],3))"; }
run arachne_amd/libarachne_amd.so X=0
run arachne_amd/variants/lib_wpe5.so X=0
run arachne_amd/variants/lib_wpe6.so X=0
run arachne_amd/variants/lib_wpe6.so ARX_SEED_BWD_BPC=24
run arachne_amd/variants/lib_wpe3.so X=0
