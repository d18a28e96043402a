#!/bin/bash
# A/B of run-time switches (and library variants: ARX_LIB=arachne_amd/variants/lib_X.so) on the default bench command, run under gpurun:
# pairs/s, ms per step and the seeding / DP kernels' times alone, per setting, same box.  Arguments: one quoted "VAR=val VAR=val" per setting.
run() { echo -n "$* : "; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end $ARX_AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['kernel_ms_per_step_alone']
print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step; parity', d['parity_ok'], '; alone:', {k: a.get(k) for k in ('seed_fwd','seed_bwd','seed_strat','locate','extend','sw_u8','chain','rfa')}, 'sum', round(sum(a.values()),1))"; }
for s in "$@"; do run $s || exit 1; done
