#!/bin/bash
# whole GPU suite, then the default command profiled (tools/gpu_profile.sh), then the other workloads' bench lines
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/gpu_profile.sh > $O/profile.log 2>&1; echo "profile rc=$?"; grep -E "sw counters|seed traffic|bench plain" $O/profile.log | cut -c1-900
for w in alt_repeat vxmix chr20; do
  timeout -k 10 600 python bench.py --workload $w --steps 5 --cpu-sample 60000 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"
done
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python3 - <<'PY'
import json
for w in ['default','alt_repeat','vxmix','chr20']:
    try:
        d=json.loads(open('gpurun_out/r03h/bench_%s.json'%w).read().strip().splitlines()[-1])
        print(w, round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms; parity', d.get('parity_ok'), 'e2e', round(d.get('end_to_end',{}).get('value',0)), 'roofline', round(d['roofline']['frac'],3), d['roofline']['bound'], 'path', round(d['roofline_path']['frac'],3), 'book', d['bookkeeping_share']['alone'])
    except Exception as e: print(w, 'ERR', e)
PY
