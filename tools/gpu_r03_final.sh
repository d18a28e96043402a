#!/bin/bash
# round 3, final artefacts: whole GPU suite, the default command profiled (tools/gpu_profile.sh), bench lines of all workloads; stops on a GPU fault
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for m in mixed repeats long; do
  timeout -k 10 300 python tools/gpu_fuzz.py 12 7100 $m > $O/fuzz_$m.log 2>&1; rc=$?; echo "fuzz $m rc=$rc: $(tail -n 1 $O/fuzz_$m.log)"
  if grep -q HSA_STATUS_ERROR $O/fuzz_$m.log; then echo "GPU FAULT in fuzz $m"; exit 1; fi
  [ $rc -eq 0 ] || exit 1
done
bash tools/gpu_profile.sh > $O/profile.log 2>&1; echo "profile rc=$?"; grep -E "sw counters|seed traffic|bench plain" $O/profile.log | cut -c1-1200
if grep -rq "HSA_STATUS_ERROR" gpurun_out/prof/*.err; then echo "GPU FAULT in the profile passes"; exit 1; fi
for w in grch38 vxmix chr20 alt_repeat; do
  extra=""; [ $w = alt_repeat ] && extra="--steps 6 --cpu-sample 60000"
  timeout -k 10 700 python bench.py --workload $w $extra > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"
  if grep -q "HSA_STATUS_ERROR\|out of memory" $O/bench_$w.err; then echo "GPU FAULT / OOM in $w"; tail -n 3 $O/bench_$w.err; exit 1; fi
done
python3 - <<'PY'
import json
for w in ['grch38','vxmix','chr20','alt_repeat']:
    try:
        d=json.loads(open('gpurun_out/r03z/bench_%s.json'%w).read().strip().splitlines()[-1])
        print(w, round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms; parity', d.get('parity_ok'), 'boundary', round(d['boundary']['value']), d['boundary'].get('matches_resident'), 'e2e', round(d.get('end_to_end',{}).get('value',0)), 'roofline', round(d['roofline']['frac'],3), d['roofline']['bound'], 'iso', round(d['roofline'].get('isolated',{}).get('frac',0),3), 'path', round(d['roofline_path']['frac'],3), 'stage', round(d['roofline_seeding_stage']['frac'],3), 'fwd', round(d['roofline_fwd']['frac'],3), 'book', d['bookkeeping_share']['alone'], 'cpu', round(d['cpu_baseline']['value']))
    except Exception as e: print(w, 'ERR', e)
PY
